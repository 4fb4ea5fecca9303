// raht_sweep.hpp -- the COARSE levels of the block loop with sub-node prediction
// (tmc3/RAHT.cpp:1306-1808 with :370-415, 503-565), one workgroup per slice,
// level after level inside ONE launch.
//
// Why.  With sub-node prediction a block waits for the reconstructed children
// of up to 12 earlier blocks of its level, and the lossy encoder also for the
// zero-run state of its predecessor in coding order (tmc3/RAHT.cpp:1618-1669).
// At the coarse levels of a frame the blocks are few and the DAG is narrow
// (1 M-point lidar frame, levels with <= 8 192 parents: 9 000 blocks, chains of
// 55..473 hops, 3..13 blocks side by side), so a level there is a serial walk
// and its time is (hops) x (cost of one hand-off).  The per-level kernel
// (raht_subnode.hpp) hands a value from wavefront to wavefront through memory:
// a write-through granule and an agent-scope poll, 2-4 us per hop with the
// loop around it -- 3.3 of the headline forward transform's 7.4 ms went into
// levels that hold 1.4 % of the frame's blocks.
//
// Here the wavefronts that walk a slice's coarse levels all sit on ONE CU, so a
// hand-off is an LDS write and an LDS read (some tens of ns):
//   * rounds of 8 blocks (8 lanes per block, as raht_subnode.hpp) are taken in
//     a fixed order -- wavefront w takes rounds w, w + W, w + 2 W, ... of the
//     slice's parents in Morton order, so every dependency belongs to a round
//     that some wavefront has already started or finished;
//   * a reconstructed child goes into a ring of the last R rounds' values in
//     LDS (value, then a per-lane flag word that names level and round; a round
//     that takes a slot over first marks its flags "claimed"; a reader loads
//     flag, value, flag: the first flag says the value was there, the second
//     that no later round had taken the slot when the value was read) and,
//     as before, into the launch's mail-box granule in memory, which serves
//     readers whose neighbour has left the ring;
//   * the zero-run words of a level live in LDS (one 32-bit word per parent);
//   * everything a level leaves for the next one (reconstruction, neighbour
//     counts, node qp) is in memory as the per-level kernels leave it, so the
//     launch can stop at any level and the per-level kernels go on from there;
//     the prepass of a level (single-child parents copied down, child
//     occupancies) runs inside the launch, between two workgroup barriers.
// The arithmetic of a block is the text of raht_subnode.hpp's loop, unchanged.
#pragma once

#include "raht_arith.hpp"
#include "raht_levels.hpp"
#include "raht_subnode.hpp"

namespace gpcc {

constexpr int kSweepMaxParents = 8192;  // parents of a slice's level the LDS tables hold
constexpr int kSweepDefaultParents = 256;  // levels the library hands to the sweep (see the header comment)

struct SweepCtx {
  int32_t li_hi;  // first (coarsest) children level of the launch
  int32_t li_lo;  // last one, inclusive
};

// zero-run words in LDS: 0 = pending, otherwise kind << 30 | (value + 1);
// kind 1 = reset-free (value = the block's first coefficient), 2 = final (value = L)
__device__ __forceinline__ uint32_t
sweep_word(int kind, int value)
{
  return ((uint32_t)kind << 30) | (uint32_t)(value + 1);
}

#ifdef GPCC_EMU
template<class T>
using LdsV = volatile T*;
#else
template<class T>
using LdsV = volatile __attribute__((address_space(3))) T*;
#endif

template<int C>
struct SweepRing {
  static constexpr int kRounds = C == 1 ? 128 : 32;  // rounds the value ring holds (a power of two)
};


// ---- the static half of a round's prologue as records --------------------------------------
// Everything of a block that the tree, the source attributes and the parameters decide alone --
// children, weights, the butterfly constants (three fixed-point Newton evaluations per pair at the
// coarse levels, where no weight is small), the normalisers, the forward transform of the source,
// the 18-neighbour search, the neighbours' child tables -- costs a wavefront ~15 us of instruction
// issue per round.  Inside the sweep that time sits in front of every round on the ONE compute unit
// that walks the slice (measured: 888 rounds of a level x 15 us / 8 wavefronts, more than the
// level's dependency chain); as a pass of its own it runs on the whole chip
// (raht_sweep_record_kernel, one 8-lane group per parent of every level the sweep takes), and a
// round of the sweep starts with 18 + 2 C coalesced loads.  Fields are lane-major: field f of lane
// t of parent j of children level li is f32[f * lanes + (rbase[li] + j) * 8 + t].
inline size_t
sweep_rec_bytes(int64_t parents, int c)
{
  return (size_t)(parents + 8) * (8 * (kSweepFields * 4 + 8 * (size_t)c) + 1) + 1024;
}

// carve the record arrays for `parents` parents out of one allocation
inline void
sweep_rec_carve(SweepRec* rec, void* mem, int64_t parents, int c)
{
  const size_t lanes = (size_t)(parents + 8) * 8;
  char* at = (char*)mem;
  rec->src = (int64_t*)at;
  at += lanes * 8 * (size_t)c;
  rec->f32 = (int32_t*)at;
  at += lanes * 4 * kSweepFields;
  rec->occ = (uint8_t*)at;
  rec->lanes = (int32_t)lanes;
}

// child occupancy of every parent of the sweep's levels
__global__ __launch_bounds__(256) void
sweep_occ_kernel(TreeView tv, SweepCtx sw, SweepRec rec)
{
  if (tree_failed(tv))
    return;
  const int li = sw.li_hi - (int)blockIdx.y;
  const int num_parents = tv.soff[li + 1][tv.num_slices];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < num_parents; j += gridDim.x * blockDim.x) {
    const int c0 = tv.fc[li + 1][j], c1 = tv.fc[li + 1][j + 1];
    uint32_t o = 0;
    for (int u = c0; u < c1; u++)
      o |= 1u << (int)(tv.key[li][u] & 7);
    rec.occ[rec.rbase[li] + j] = (uint8_t)o;
  }
}

template<int C, bool ENC, class A = ArithI64>
__global__ __launch_bounds__(256) void
raht_sweep_record_kernel(LevelCtx ctx, SweepCtx sw, SweepRec rec)
{
  typedef typename A::T VT;
  typedef typename A::Coef VC;
  constexpr bool kEnc = ENC;
  __shared__ SharedLut lut_s;
  if (tree_failed(ctx.tv))
    return;
  load_lut(&lut_s, ctx.lut);
  const SharedLut& lut = lut_s;
  const TreeView& tv = ctx.tv;
  const ParamsConst prm = (ParamsConst)ctx.params;
  const int li = sw.li_hi - (int)blockIdx.y;
  const int t = threadIdx.x & 7;
  const int lane = lane_id();
  const bool ext = A::kF64 || prm->raht_extension != 0;
  bool in_range = true;
  const bool by_list = rec.worklist != nullptr;
  const int num_parents = by_list ? rec.work_count[li] : tv.soff[li + 1][tv.num_slices];
  const int wave_global = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  for (int j8 = wave_global * 8; j8 < num_parents; j8 += (int)gridDim.x * 32) {
    const int jj = j8 + (lane >> 3);
    const bool live0 = jj < num_parents;
    const int j = live0 ? (by_list ? rec.worklist[jj] : jj) : 0;
    const int s = find_slice(tv.soff[li + 1], tv.num_slices, j);
    const LevelSched e = ctx.sched[s].lvl[li];
    const bool live = live0 && e.processed;
    const int sp0 = tv.soff[li + 1][s];
    const int sp1 = tv.soff[li + 1][s + 1];
    const bool inherit_dc = !e.is_root;
    const int c0 = live ? tv.fc[li + 1][j] : 0;
    const int nchild = live ? tv.fc[li + 1][j + 1] - c0 : 0;
    // all shuffles below run in wave-uniform control flow; lanes of dead groups carry zeros
    const bool on = live && !(ext && nchild == 1);
    {
      // ---- children -> positions ---------------------------------------
      const int64_t ckey = (on && t < nchild) ? tv.key[li][c0 + t] : 0;
      const uint32_t occ = group8_or((on && t < nchild) ? 1u << (int)(ckey & 7) : 0u);
      const bool has = (occ >> t) & 1;
      const int child = c0 + popc32(occ & ((1u << t) - 1));
      int32_t w = 0;
      VT src[C];
#pragma unroll
      for (int k = 0; k < C; k++)
        src[k] = A::zero();
      if (has) {
        const int f0 = tv.fp[li][child], f1 = tv.fp[li][child + 1];
        w = f1 - f0;
        if (kEnc) {
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] = A::from_int((int32_t)(
              (uint32_t)ctx.attr_prefix[(size_t)f1 * C + k]
              - (uint32_t)ctx.attr_prefix[(size_t)f0 * C + k]));
        }
      }

      // ---- butterfly weights + coefficients (mkWeightTree :742) ----------
      int32_t wl[3], wr[3];
      VC ca[3], cb[3];
      int32_t ia3[3], ib3[3];
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
          raht_coeffs(wl[st], wr[st], lut, &ia, &ib);
          cw = wl[st] + wr[st];
        } else {
          cw = left ? wl[st] + wr[st] : 0;
        }
        ca[st] = A::coef(ia);
        cb[st] = A::coef(ib);
        ia3[st] = (int32_t)ia;
        ib3[st] = (int32_t)ib;
      }

      // ---- the 18-neighbour search (findNeighbours, tmc3/RAHT.cpp:299-368), wherever the level predicts:
      //      whether the parent's own neighbour count lets the block predict is the sweep's to say ----
      const bool do_search = on && inherit_dc && prm->raht_prediction_enabled_flag != 0;
      int pn[3] = {-1, -1, -1};  // neighbour i = 1 + t + 8*slot
      {
        // the three lower_bound searches of a lane advance in lock step, so
        // their probes are in flight together (12 dependent steps, not 36)
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, end[3] = {0, 0, 0};
        int64_t want[3] = {0, 0, 0};
        if (do_search) {
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
      int found_sum = (pn[0] >= 0) + (pn[1] >= 0) + (pn[2] >= 0);
      found_sum = group8_sum(found_sum);
      const uint32_t present = group8_bits(on && cw != 0) | (on ? 1u : 0u);
      // the two normalisers of this position, settled before the wait
      int32_t nrm_sq_i = 0, nrm_rs_i = 0, nrm_shift = 0;
      if (w > 1) {
        nrm_sq_i = (int32_t)sqrt_weight(w, lut);
        if (w < kSmallN) {
          nrm_rs_i = lut.norm_rs[w];
        } else {
          const uint64_t w64 = (uint64_t)w;
          nrm_shift = w64 > 1024 ? ilog2_u64(w64 - 1) >> 1 : 0;
          nrm_rs_i = (int32_t)(irsqrt(w64, lut.rsqrt) >> (40 - nrm_shift - kFpFrac));
        }
      }
      const VC nrm_rs = A::coef(nrm_rs_i);
      if (kEnc) {
        // forward butterflies of the source (normalised first: scale_rsqrt, tmc3/RAHT.cpp:1474-1481)
        if (w > 1) {
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] = A::mulc(A::shr(src[k], nrm_shift), nrm_rs);
        }
#pragma unroll
        for (int k = 0; k < C; k++)
          in_range = in_range && A::below(src[k], A::kFwdLimit);
#pragma unroll
        for (int st = 0; st < 3; st++) {
          const int bit = 1 << st;
          const bool left = !(t & bit);
          const bool both = wl[st] && wr[st];
          const bool swap = !wl[st] && wr[st];
#pragma unroll
          for (int k = 0; k < C; k++) {
            const VT own = src[k], oth = shfl_xor_v(own, bit);
            if (both) {
              src[k] = left ? A::mulc(oth, cb[st]) + A::mulc(own, ca[st])
                            : A::mulc(own, ca[st]) - A::mulc(oth, cb[st]);
            } else if (swap) {
              src[k] = oth;
            }
          }
        }
      }
      // ---- the neighbours this lane owns: first child, occupancy, single-child bit -------------
      int nb_c0[3] = {0, 0, 0};
      uint32_t nb_occ[3] = {0, 0, 0};
      int nb_single[3] = {0, 0, 0};
#pragma unroll
      for (int slot = 0; slot < 3; slot++) {
        const int i = 1 + t + 8 * slot;
        if (do_search && i >= 7 && i < 19 && pn[slot] >= 0) {
          const int q = pn[slot];
          if (q < j) {  // processed before this block: its children count
            const int qc0 = tv.fc[li + 1][q];
            nb_c0[slot] = qc0;
            nb_occ[slot] = rec.occ[rec.rbase[li] + q];
            nb_single[slot] = ext && tv.fc[li + 1][q + 1] - qc0 == 1;
          }
        }
      }
      uint32_t bothm = 0, swapm = 0;
#pragma unroll
      for (int st = 0; st < 3; st++) {
        bothm |= (wl[st] && wr[st]) ? 1u << st : 0u;
        swapm |= (!wl[st] && wr[st]) ? 1u << st : 0u;
      }
      if (live0) {
        const size_t at = (size_t)(by_list ? jj : rec.rbase[li] + j) * 8 + t;
        int32_t* __restrict__ f = rec.f32 + at;
        const size_t ln = (size_t)rec.lanes;
        f[kSfW * ln] = w;
#pragma unroll
        for (int st = 0; st < 3; st++) {
          f[(kSfCa + st) * ln] = ia3[st];
          f[(kSfCb + st) * ln] = ib3[st];
          f[(kSfPn + st) * ln] = pn[st];
          f[(kSfNbc0 + st) * ln] = nb_c0[st];
        }
        f[kSfNsq * ln] = nrm_sq_i;
        f[kSfNrs * ln] = nrm_rs_i;
        f[kSfPk * ln] = (int32_t)(nb_occ[0] | nb_occ[1] << 8 | nb_occ[2] << 16
                                  | (uint32_t)(nb_single[0] | nb_single[1] << 1 | nb_single[2] << 2) << 24
                                  | (uint32_t)nrm_shift << 27);
        f[kSfPk2 * ln] = (int32_t)(occ | present << 8 | (uint32_t)found_sum << 16 | bothm << 21 | swapm << 24
                                   | (on ? 1u << 27 : 0u));
        f[kSfC0 * ln] = c0;
        f[kSfSlice * ln] = s;
        if (kEnc) {
#pragma unroll
          for (int k = 0; k < C; k++)
            rec.src[(size_t)k * ln + at] = __builtin_bit_cast(int64_t, src[k]);
        }
      }
    }
  }
  if (A::kF64 && __any(!in_range) && lane == 0)
    atomicCAS(ctx.error, 0, 3);
}

template<int C, int MODE, class A = ArithI64, int NT = 512>
__global__ __launch_bounds__(NT) void
raht_sub_sweep_kernel(LevelCtx ctx, SweepCtx sw, SweepRec rec)
{
  static_assert(MODE == kSynth || MODE == kLossySub, "mode");
  static_assert(NT % 64 == 0 && NT <= 1024, "whole wavefronts");
  typedef typename A::T VT;
  typedef typename A::Coef VC;
  constexpr bool kLossy = MODE == kLossySub;
  constexpr bool kEnc = MODE != kSynth;
  constexpr int NW = NT / 64;
  constexpr int R = SweepRing<C>::kRounds;
  static_assert(R % NW == 0, "a ring slot belongs to one wavefront");

  __shared__ SharedLut lut_s;
  __shared__ unsigned long long ring_val_s[R * 64 * C];
  __shared__ int32_t ring_flag_s[R * 64];
  __shared__ uint32_t rs_s[kLossy ? kSweepMaxParents : 1];
  __shared__ int32_t l_cur_s;  // last reset of the slice so far (tmc3/RAHT.cpp:1618-1669 across levels)
  // (volatile accesses through LDS-qualified pointers: ds_read / ds_write in program order; through a generic
  // pointer they would be system-scope FLAT instructions)
  const LdsV<unsigned long long> ring_val = (LdsV<unsigned long long>)ring_val_s;
  const LdsV<int32_t> ring_flag = (LdsV<int32_t>)ring_flag_s;
  const LdsV<uint32_t> rs = (LdsV<uint32_t>)rs_s;

  if (tree_failed(ctx.tv))
    return;
  load_lut(&lut_s, ctx.lut);
  const SharedLut& lut = lut_s;

  const TreeView& tv = ctx.tv;
  const ParamsConst prm = (ParamsConst)ctx.params;
  const int s = blockIdx.x;  // the slice of this workgroup
  const int t = threadIdx.x & 7;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int gbase = threadIdx.x & 56;  // first lane of this 8-lane group
  const bool ext = A::kF64 || prm->raht_extension != 0;
  int pwc12[12];
#pragma unroll
  for (int i12 = 0; i12 < 12; i12++)
    pwc12[i12] = prm->pred_weight_child[i12];
  bool in_range = true;  // (ArithF64: the magnitudes that bound every product, raht_arith.hpp)
  const int pt0 = tv.pt_off[s];
  const int n_s = tv.pt_off[s + 1] - pt0;
  const auto mrsrc = __builtin_amdgcn_make_buffer_rsrc(
    ctx.mbox, 0, (int)((size_t)tv.n_total * C * 16), 0x00020000);

  for (int i = threadIdx.x; i < R * 64; i += NT)
    ring_flag_s[i] = 0;
  if (kLossy && threadIdx.x == 0)
    l_cur_s = ctx.slice_l[((sw.li_hi + 1) & 1) * tv.num_slices + s];
  __syncthreads();

  for (int li = sw.li_hi; li >= sw.li_lo; li--) {
    const LevelSched e = ctx.sched[s].lvl[li];
    if (!e.processed)
      continue;  // (workgroup-uniform)
    const int32_t seq = sw.li_hi - li + 1;  // levels of the launch in order: ring flags only grow
    const uint32_t mtag = (uint32_t)(li + 1);
    const int sp0 = tv.soff[li + 1][s];  // slice's parents
    const int sp1 = tv.soff[li + 1][s + 1];
    const int sc0 = tv.soff[li][s];      // slice's children
    const int P = sp1 - sp0;
    const int par_par = e.parity ^ 1, cur_par = e.parity;
    const bool inherit_dc = !e.is_root;
    const int l_in = kLossy ? l_cur_s : -1;  // the state the previous level left
    if (P > kSweepMaxParents) {
      // (the host sizes the launch from the tree's statistics; never silently wrong)
      if (threadIdx.x == 0)
        atomicExch(ctx.error, 1);
      return;
    }

    // ---- the level's prepass (raht_levels.hpp, raht_level_prepass_kernel): child occupancies,
    //      single-child parents finished on the spot, their zero-run words -------------------------
    for (int p = threadIdx.x; p < P; p += NT) {
      const int j = sp0 + p;
      const int c0 = tv.fc[li + 1][j];
      const int nchild = tv.fc[li + 1][j + 1] - c0;
      if (kLossy)
        rs_s[p] = 0;
      if (ext && nchild == 1) {
        const int64_t prow = (int64_t)pt0 + p;
        const int64_t crow = (int64_t)pt0 + (c0 - sc0);
#pragma unroll
        for (int k = 0; k < C; k++) {
          par2(ctx.rec_us, cur_par)[crow * C + k] = par2(ctx.rec_us, par_par)[prow * C + k];
          par2(ctx.rec, cur_par)[crow * C + k] = par2(ctx.rec, par_par)[prow * C + k];
        }
        par2(ctx.nneigh, cur_par)[crow] = 19;
        if (kLossy)  // no coefficient: reset-free, the next block's first coefficient is its own
          rs_s[p] = sweep_word(1, e.coeff_base + (inherit_dc ? (c0 - sc0) - p : 0));
      }
    }
    // what the prepass stored is read with ordinary loads by the rounds below: stores complete, barrier,
    // and only then the CU's L1 is invalidated (a wavefront still in front of the barrier could refill it)
    __threadfence();
    __syncthreads();
    __threadfence();

    // quantisers of this lane's coefficient position: without region QPs (the host keeps those calls on the
    // per-level kernels) they depend on the level and the position alone
    Quantizer qa[2] = {{1, 1}, {1, 1}};
    Quantizer qr[2] = {{1, 1}, {1, 1}};
    bool q_same = true;  // no AC offset here: the RDOQ quantiser is the coding quantiser
    {
      int ac0 = 0, ac1 = 0;
      if (e.ac_layer < prm->num_ac_qp_layers && t) {
        ac0 = prm->ac_qp_offset[e.ac_layer][t - 1][0];
        ac1 = prm->ac_qp_offset[e.ac_layer][t - 1][1];
        q_same = (ac0 | ac1) == 0;
      }
      qpset_quantizers(prm, e.qp_layer, ac0, ac1, qa);
      if (kLossy)
        qpset_quantizers(prm, e.qp_layer, 0, 0, qr);
    }
    const typename A::Quant qaa[2] = {A::quant(qa[0]), A::quant(qa[1])};
    const typename A::Quant qra[2] = {A::quant(qr[0]), A::quant(qr[1])};
    const size_t ln = (size_t)rec.lanes;

    for (int r = wave; r * 8 < P; r += NW) {
      // a bounded wait has expired somewhere: the result is discarded anyway
      if (__hip_atomic_load(ctx.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        break;
      SubProf prof;
      prof.round_begin();
      const int rslot = r & (R - 1);
      const int32_t my_flag = (seq << 20) | r;
      ring_flag[rslot * 64 + lane] = my_flag | (int32_t)0x80000000;  // claimed, values not there yet
      const int p = r * 8 + (lane >> 3);
      const bool live = p < P;
      const int j = sp0 + (live ? p : 0);
      // ---- the block's record (raht_sweep_record_kernel) ---------------------
      const size_t rat = (size_t)(rec.rbase[li] + sp0 + r * 8) * 8 + lane;
      const int32_t* __restrict__ rf = rec.f32 + rat;
      const uint32_t pk2 = live ? (uint32_t)rf[kSfPk2 * ln] : 0u;
      const uint32_t pk = (uint32_t)rf[kSfPk * ln];
      const int c0 = rf[kSfC0 * ln];
      const int32_t w = live ? rf[kSfW * ln] : 0;
      const int32_t nrm_sq_i = rf[kSfNsq * ln], nrm_rs_i = rf[kSfNrs * ln];
      int pn[3], nb_c0[3];
      VC ca[3], cb[3];
#pragma unroll
      for (int st = 0; st < 3; st++) {
        ca[st] = A::coef(rf[(kSfCa + st) * ln]);
        cb[st] = A::coef(rf[(kSfCb + st) * ln]);
        pn[st] = rf[(kSfPn + st) * ln];
        nb_c0[st] = rf[(kSfNbc0 + st) * ln];
      }
      VT src[C];
#pragma unroll
      for (int k = 0; k < C; k++)
        src[k] = kEnc ? __builtin_bit_cast(VT, rec.src[(size_t)k * ln + rat]) : A::zero();
      // all shuffles below run in wave-uniform control flow; lanes of dead
      // groups carry zeros and store nothing
      const uint32_t occ = pk2 & 0xffu;
      const uint32_t present = (pk2 >> 8) & 0xffu;
      const int found_sum = (int)((pk2 >> 16) & 31u);
      const uint32_t bothm = (pk2 >> 21) & 7u, swapm = (pk2 >> 24) & 7u;
      const bool on = (pk2 >> 27) & 1u;
      const int nrm_shift = (int)(pk >> 27);
      const uint32_t nb_occ[3] = {pk & 0xffu, (pk >> 8) & 0xffu, (pk >> 16) & 0xffu};
      const int nb_single[3] = {(int)((pk >> 24) & 1u), (int)((pk >> 25) & 1u), (int)((pk >> 26) & 1u)};
      const int pj = p;
      const int64_t prow = (int64_t)pt0 + pj;  // parent row in rec buffers
      const bool has = (occ >> t) & 1;
      const int child = c0 + popc32(occ & ((1u << t) - 1));
      const int64_t crow = (int64_t)pt0 + (child - sc0);

      // ---- inter-level prediction (tmc3/RAHT.cpp:1391-1432) --------------
      const bool pred_in_level = on && inherit_dc && prm->raht_prediction_enabled_flag != 0;
      bool enable_pred = pred_in_level;
      int neigh_count = 0;
      VT pred[C];
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] = A::zero();
      bool do_search = false;
      if (pred_in_level) {
        if (par2(ctx.nneigh, par_par)[prow] < prm->raht_prediction_threshold0)
          enable_pred = false;
        else
          do_search = true;
      }
      if (do_search) {
        neigh_count = found_sum + 1;
        if (neigh_count < prm->raht_prediction_threshold1)
          enable_pred = false;
      }

      // ---- coefficient slot of this position (scanBlock :776-791) --------
      // scan order 0,4,2,1,6,5,3,7 -> scan position of t
      const int spos = (0x74516230u >> (4 * t)) & 7;
      const uint32_t pscan = ((present >> 0) & 1) | (((present >> 4) & 1) << 1)
        | (((present >> 2) & 1) << 2) | (((present >> 1) & 1) << 3)
        | (((present >> 6) & 1) << 4) | (((present >> 5) & 1) << 5)
        | (((present >> 3) & 1) << 6) | (((present >> 7) & 1) << 7);
      const int rank = popc32(pscan & ((1u << spos) - 1));
      const bool coded = on && ((present >> t) & 1) && (t != 0 || !inherit_dc);
      // slice-relative coefficient index
      const int cidx = e.coeff_base + (inherit_dc ? (c0 - sc0) - pj + rank - 1 : rank);
      int32_t* __restrict__ cplane = ctx.coeffs + (size_t)pt0 * C + cidx;

      const VC nrm_sq = A::coef(nrm_sq_i), nrm_rs = A::coef(nrm_rs_i);
      // RDOQ bookkeeping of the lossy encoder: rank of this lane's coefficient
      // among the block's coded coefficients, first coefficient index
      const uint32_t coded_mask = group8_bits(coded);
      const int ncoef = popc32(coded_mask);
      const int crank = inherit_dc ? rank - 1 : rank;         // valid when coded
      const int cfirst = e.coeff_base + (inherit_dc ? (c0 - sc0) - pj : 0);
      int rank_src = 0;  // lane of the group whose coefficient has rank t (t < ncoef)
      if (kLossy) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int ru = __shfl(crank, gbase | u);
          if (((coded_mask >> u) & 1) && ru == t)
            rank_src = u;
        }
      }
      bool lin_known = false;
      int lin = -1;
      VT dc[C];
#pragma unroll
      for (int k = 0; k < C; k++)
        dc[k] = A::zero();
      if (on && inherit_dc && t == 0) {
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int64_t val = par2(ctx.rec_us, par_par)[prow * C + k];
          dc[k] = A::from_i64(
            ext ? val : (val > 0 ? val << (kFpFrac - 2) : -((-val) << (kFpFrac - 2))));
          in_range = in_range && A::below(dc[k], A::kInvLimit);
        }
      }

      const bool run = do_search && enable_pred;
      const int64_t* __restrict__ prec = par2(ctx.rec, par_par);
      const int64_t rbase = (int64_t)pt0 - sp0;
      int wsum = 0;
      VT lim_lo = A::zero(), lim_hi = A::zero();
      // intraDcPred, the seven neighbours that never use child values
      // (tmc3/RAHT.cpp:463-502 with parentOnlyCheckMaxIdx = 7)
#pragma unroll
      for (int i = 0; i < 7; i++) {
        int q;
        if (i == 0)
          q = j;
        else
          q = __shfl(pn[0], gbase | (i - 1));
        if (!run || q < 0)
          continue;
        VT v[C];
#pragma unroll
        for (int k = 0; k < C; k++) {
          v[k] = A::from_i64(prec[(rbase + q) * C + k]);
          in_range = in_range && A::below(v[k], A::kRecLimit);
        }
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

      // ---- neighbours 7..18 (tmc3/RAHT.cpp:503-565): everything that does
      //      not depend on this level's rounds is settled here ------------------
      // lane t owns neighbours i = 1 + t + 8*slot (the lanes that searched them)
      VT nb_v[3][C];
#pragma unroll
      for (int slot = 0; slot < 3; slot++) {
#pragma unroll
        for (int k = 0; k < C; k++)
          nb_v[slot][k] = A::zero();
        const int i = 1 + t + 8 * slot;
        if (run && i >= 7 && i < 19 && pn[slot] >= 0) {
          const int q = pn[slot];
#pragma unroll
          for (int k = 0; k < C; k++) {
            nb_v[slot][k] = A::from_i64(prec[(rbase + q) * C + k]);
            in_range = in_range && A::below(nb_v[slot][k], A::kRecLimit);
          }
        }
      }
      uint32_t pend = 0;       // neighbours whose child is awaited
      int32_t nrow12[12];      // row of that child in rec / mbox
      int32_t wq12[12];        // its place in the ring: round << 6 | lane
      // inw: awaited children that may still be in the LDS ring
      uint32_t inw = 0;
#pragma unroll
      for (int i12 = 0; i12 < 12; i12++) {
        nrow12[i12] = 0;
        wq12[i12] = 0;
        const int i = 7 + i12;
        const int owner = gbase | ((i - 1) & 7);
        const int sl = (i - 1) >> 3;
        const int q = __shfl(pn[sl], owner);
        const int qc0 = __shfl(nb_c0[sl], owner);
        const uint32_t qocc = __shfl(nb_occ[sl], owner);
        const int single = __shfl(nb_single[sl], owner);
        VT v[C];
#pragma unroll
        for (int k = 0; k < C; k++)
          v[k] = shfl_v(nb_v[sl][k], owner);
        if (!run || q < 0)
          continue;
        if (A::muli(v[0], 10) <= lim_lo || A::muli(v[0], 10) >= lim_hi)
          continue;
        if (has && ((neigh_mask(i) >> t) & 1)) {
          const int sh = occu_shift(i12);
          const int cpos = i12 < 9 ? t + sh : t - sh;
          const bool child_ok = cpos >= 0 && cpos < 8 && ((qocc >> cpos) & 1);
          if (child_ok) {
            const int cidx_n = qc0 + popc32(qocc & ((1u << cpos) - 1));
            const int64_t nrow = (int64_t)pt0 + (cidx_n - sc0);
            const int pwc = prm->pred_weight_child[i12];
            wsum += pwc;
            if (single) {
              // copied by the level's prepass: an ordinary load
              const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
              for (int k = 0; k < C; k++) {
                const VT cv = A::from_i64(par2(ctx.rec, cur_par)[nrow * C + k]);
                in_range = in_range && A::below(cv, A::kRecLimit);
                pred[k] += A::muli(cv, mul);
              }
            } else {
              nrow12[i12] = (int32_t)nrow;
              const int wq = ((q - sp0) << 3) | cpos;
              wq12[i12] = wq;
              pend |= 1u << i12;
              if ((wq >> 6) > r - R)
                inw |= 1u << i12;
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
      const VC pdiv = A::coef(pred_divisor(wsum > 0 ? wsum : 1));

      // ---- awaited children that have left the ring: their rounds are long done, so ALL their granules are
      //      fetched here side by side (the loop below polls one granule per lane and iteration -- a memory
      //      round trip each); whatever is not there yet stays awaited --------------------------------------
      {
        constexpr int kBatch = C == 1 ? 12 : 4;
#pragma unroll
        for (int b0 = 0; b0 < 12; b0 += kBatch) {
          const uint32_t far = pend & ~inw;
          if (__any(((far >> b0) & ((1u << kBatch) - 1u)) != 0)) {
            u32x4 g[kBatch][C];
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
#pragma unroll
              for (int k = 0; k < C; k++) {
                g[u][k] = u32x4{0u, 0u, 0u, 0u};
                if ((far >> (b0 + u)) & 1u)
                  g[u][k] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, (nrow12[b0 + u] * C + k) * 16, 0, /*sc1*/ 16);
              }
            }
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
              bool ok = (far >> (b0 + u)) & 1u;
#pragma unroll
              for (int k = 0; k < C; k++)
                ok = ok && g[u][k].z == mtag;
              if (ok) {
                const int pwc = pwc12[b0 + u];
                const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
                for (int k = 0; k < C; k++)
                  pred[k] += A::muli(__builtin_bit_cast(VT, ((uint64_t)g[u][k].y << 32) | g[u][k].x), mul);
                pend &= ~(1u << (b0 + u));
              }
            }
          }
        }
      }

      // ---- the staged dependency loop (raht_subnode.hpp) ---------------------
      // stage 0: waiting for neighbour blocks   -> (P) predict + transform
      // stage 1: waiting for the RDOQ state L (lossy encoder only)
      // stage 3: committed
      int stage = on ? 0 : 3;
      unsigned spins = 0;
      VT pt[C];               // transformed prediction of this position
      int32_t qc[C];          // tentative quantised coefficients (encoder) / coded ones (decoder)
      uint32_t dr = kDescZero;  // RDOQ descriptor of rank t (lossy encoder)
      int zs = 0;
      bool zero_r = false;
      uint32_t dmask = 0;     // definite resets of the block, by rank
      bool lin_exact = false;
      int need = 0;           // reach of the block's thresholds before its first coefficient
      int look = p - 1;       // look-back cursor (slice-local parent index)
      // outgoing RDOQ state: 0 unknown, 1 transparent, 2 final (= outv); a parent the prepass
      // finished has no coefficient
      int outk = (live && !on) ? 1 : 0, outv = -1;
#pragma unroll
      for (int k = 0; k < C; k++) {
        pt[k] = A::zero();
        // decoder: the coded coefficients are input -- fetched here, not on the chain
        qc[k] = (!kEnc && coded) ? cplane[(size_t)k * n_s] : 0;
      }
      int probe_wq = -1;  // the awaited ring child latest in Morton order
#pragma unroll
      for (int i12 = 0; i12 < 12; i12++)
        probe_wq = (((pend & inw) >> i12) & 1) && wq12[i12] > probe_wq ? wq12[i12] : probe_wq;
      prof.loop_begin();
      while (__any(stage != 3)) {
        bool progressed = false;
        prof.iter_begin();
        // ---- (X) awaited children that may be in the ring ------------------------
        // A waiting lane looks at ONE flag per iteration: that of its awaited child latest in Morton order
        // (probe_wq) -- the rounds commit roughly in order, so when that one is there the others are.  Only
        // when some lane's probe has arrived (or its slot has a new owner) does the wavefront go through all
        // awaited slots: flag, value, flag again -- the first read says the value was written before, the
        // second that no later round had taken the slot over when the value was read.
        {
          const uint32_t pin = stage == 0 ? (pend & inw) : 0u;
          bool hit = false;
          if (pin) {
            const int rq = probe_wq >> 6;
            const int32_t fl = ring_flag[(rq & (R - 1)) * 64 + (probe_wq & 63)];
            const int32_t wantf = (seq << 20) | rq;
            hit = fl == wantf || (fl & 0x7fffffff) > wantf;
          }
          if (__any(hit)) {
            uint32_t mi = pin;
            while (__any(mi != 0)) {
              const bool act = mi != 0;
              const int slot = act ? __ffs(mi) - 1 : 0;
              mi &= mi - 1;
              int wq = 0, pwc = 0;
#pragma unroll
              for (int i12 = 0; i12 < 12; i12++) {
                wq = slot == i12 ? wq12[i12] : wq;
                pwc = slot == i12 ? pwc12[i12] : pwc;
              }
              const int rq = wq >> 6;
              const int ra = (rq & (R - 1)) * 64 + (wq & 63);
              const int32_t fl = ring_flag[ra];
              unsigned long long vb[C];
#pragma unroll
              for (int k = 0; k < C; k++)
                vb[k] = ring_val[ra * C + k];
              const int32_t fl2 = ring_flag[ra];
              const int32_t wantf = (seq << 20) | rq;
              if (act && fl == wantf && fl2 == wantf) {
                const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
                for (int k = 0; k < C; k++)
                  pred[k] += A::muli(__builtin_bit_cast(VT, vb[k]), mul);
                pend &= ~(1u << slot);
                inw &= ~(1u << slot);
              } else if (act && (fl2 & 0x7fffffff) > wantf) {
                inw &= ~(1u << slot);  // the slot has a new owner: the granule in memory serves
              }
            }
            // the next probe: the awaited ring child latest in Morton order
            probe_wq = -1;
#pragma unroll
            for (int i12 = 0; i12 < 12; i12++)
              probe_wq = (((pend & inw) >> i12) & 1) && wq12[i12] > probe_wq ? wq12[i12] : probe_wq;
            progressed = true;
          }
        }
        // ---- (X) the others: the granule is data and flag at once -----------
        const uint32_t pm = stage == 0 ? (pend & ~inw) : 0u;
        if (pm) {
          const int slot = __ffs(pm) - 1;
          int32_t row = 0;
          int pwc = 0;
#pragma unroll
          for (int i12 = 0; i12 < 12; i12++) {
            if (slot == i12) {
              row = nrow12[i12];
              pwc = pwc12[i12];
            }
          }
          u32x4 g[C];
#pragma unroll
          for (int k = 0; k < C; k++)
            g[k] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, (row * C + k) * 16, 0, /*sc1*/ 16);
          bool ok = true;
#pragma unroll
          for (int k = 0; k < C; k++)
            ok = ok && g[k].z == mtag;
          if (ok) {
            const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
            for (int k = 0; k < C; k++)
              pred[k] += A::muli(__builtin_bit_cast(VT, ((uint64_t)g[k].y << 32) | g[k].x), mul);
            pend &= ~(1u << slot);
            progressed = true;
          }
        }
        prof.mark<0>();
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
            for (int k = 0; k < C; k++)
              pw_[k] = A::mulc(pw_[k], pdiv);
          }
          if (w > 1 && enable_pred) {
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
            const bool both = (bothm >> st) & 1u;
            const bool swap = (swapm >> st) & 1u;
#pragma unroll
            for (int k = 0; k < C; k++) {
              const VT own = pw_[k], oth = shfl_xor_v(own, bit);
              if (enable_pred) {
                if (both) {
                  pw_[k] = left ? A::mulc(oth, cb[st]) + A::mulc(own, ca[st])
                                : A::mulc(own, ca[st]) - A::mulc(oth, cb[st]);
                } else if (swap) {
                  pw_[k] = oth;
                }
              }
            }
          }
          // encoder: residual, tentative coefficient, RDOQ descriptor
          uint32_t d = kDescZero;
          int32_t qn_[C];
#pragma unroll
          for (int k = 0; k < C; k++)
            qn_[k] = 0;
          if (kEnc && coded) {
            int64_t sum_coeff = 0, dist2 = 0;
            int rate_coeff = 0;
#pragma unroll
            for (int k = 0; k < C; k++) {
              const VT res = enable_pred ? src[k] - pw_[k] : src[k];
              const VT co = A::round_int(res);
              qn_[k] = A::quantize(qaa[k ? 1 : 0], co);
              if (kLossy) {
                const int64_t coi = A::to_small(co);  // (|co| < 2^21 inside the checked range)
                dist2 += coi * coi;
                int64_t aq = q_same ? (int64_t)qn_[k] : (int64_t)A::quantize(qra[k ? 1 : 0], co);
                aq = aq < 0 ? -aq : aq;
                sum_coeff += aq;
                rate_coeff += rate_log_small(aq);
              }
            }
            if (kLossy) {
              d = kDescNever;
              if (sum_coeff < 3) {
                // (an all-zero coefficient that no AC-offset quantiser made non-zero never looks at its threshold:
                // raht_subnode.hpp)
                bool any = false;
#pragma unroll
                for (int k = 0; k < C; k++)
                  any |= qn_[k] != 0;
                if (sum_coeff == 0 && !any) {
                  d = kDescZero;
                } else {
                  const int64_t l0 = qr[0].step;
                  d = rdoq_threshold(dist2, l0 * l0 * (C == 1 ? 25 : 35), rate_coeff, (uint32_t)n_s);
                  if (sum_coeff == 0)
                    d |= kDescZero;
                }
              }
            }
          }
          uint32_t drn = kDescZero;
          if (kLossy) {
            // descriptors in coding order: lane r of the group gets rank r
            const uint32_t du = __shfl(d, gbase | rank_src);
            drn = t < ncoef ? du : kDescZero;
          }
          if (nready) {
#pragma unroll
            for (int k = 0; k < C; k++) {
              pt[k] = pw_[k];
              if (kEnc)
                qc[k] = qn_[k];
            }
            dr = drn;
            stage = 1;
          }
        }

        // ---- (Z) RDOQ state: can the stage-1 groups commit? ----------------
        // (tmc3/RAHT.cpp:1618-1669 restated in raht_rdoq.hpp, the walk of raht_subnode.hpp with
        // the words of the level in LDS)
        prof.mark<1>();
        bool can = stage == 1;
        if (kLossy) {
          const bool rvalid = t < ncoef;
          const bool rz = dr >> 31;
          const uint32_t rthr = dr & kDescNever;
          const bool isthr = rvalid && !rz && rthr != kDescNever && rthr != 0;
          const bool dep = rvalid && rthr != kDescNever && rthr != 0;  // its zeroing depends on the run
          const int ci = cfirst + t;  // slice-relative index of rank t
          const uint32_t below = (1u << t) - 1u;
          auto resets_for = [&](int l0, bool act) -> uint32_t {
            uint32_t resets = dmask;
            for (;;) {
              const uint32_t bb = resets & below;
              const int lhat = bb ? cfirst + (31 - __clz(bb)) : l0;
              const bool fail = act && isthr && !((resets >> t) & 1) && (uint32_t)(ci - lhat) <= rthr;
              const unsigned long long m = __ballot(fail);
              if (!m)
                break;
              resets |= (uint32_t)(m >> gbase) & 0xffu;
            }
            return resets;
          };
          auto zeroed = [&](uint32_t resets, int l0) -> bool {
            const uint32_t bb = resets & below;
            const int lhat = bb ? cfirst + (31 - __clz(bb)) : l0;
            return rvalid && rthr != kDescNever && (uint32_t)(ci - 1 - lhat) >= rthr;
          };
          // a block that has just got its descriptors
          const bool entry = stage == 1 && zs == 0;
          if (__any(entry)) {
            const bool isdef = rvalid && !rz && rthr == kDescNever;
            const uint32_t dm = group8_bits(entry && isdef);
            const bool tany = group8_any(entry && dep);
            if (entry) {
              dmask = dm;
              zs = 1;
              if (cfirst == e.coeff_base) {
                // the slice's first coded block of the level: the state the previous level left
                lin = l_in;
                lin_known = lin_exact = true;
              } else if (!tany) {
                // no coefficient looks at the run: decided, and what the block does to L is known
                zero_r = rvalid && rthr == 0;
                zs = 2;
                outk = dm ? 2 : 1;
                outv = dm ? cfirst + (31 - __clz(dm)) : outv;
              }
            }
            // the others: the two extreme hypotheses for the incoming L, once
            const bool hyp = entry && zs == 1 && !lin_known;
            if (__any(hyp)) {
              const uint32_t ra = resets_for(-1, hyp);
              const uint32_t rb = resets_for(cfirst - 1, hyp);
              const bool fa = zeroed(ra, -1), fb = zeroed(rb, cfirst - 1);
              const bool same = !group8_any(hyp && fa != fb) && ra == rb;
              const int la = ra ? 31 - __clz(ra) : -1, lb = rb ? 31 - __clz(rb) : -1;
              const int nd = group8_max((hyp && dep) ? (int)rthr - t : 0);
              if (hyp) {
                need = nd;
                if (same) {
                  zero_r = fb;
                  zs = 2;
                  outk = ra ? 2 : 1;
                  outv = ra ? cfirst + la : outv;
                } else if (ra && la == lb) {
                  // decisions still open, outgoing L already certain: successors go on
                  outk = 2;
                  outv = cfirst + lb;
                  if (t == 0)
                    rs[p] = sweep_word(2, outv);
                }
              }
            }
          }
          if (__any(stage == 1 && zs == 1)) {
            // ---- the incoming L: predecessors' words (LDS), 8 per step; only a reset within `need`
            // coefficients before the block matters ----
            const bool wantm = stage == 1 && zs == 1 && !lin_known;
            if (__any(wantm)) {
              const int kk = look - t;
              const int kc = kk < 0 ? 0 : kk;
              uint32_t sv = 0;
              if (wantm)
                sv = rs[kc];
              const bool boundary = kk < 0;
              // kind: 0 pending, 1 reset-free (value = its first coefficient),
              // 2 final (value = L), 3 slice start, 5 far enough
              int kind = boundary ? 3 : (int)(sv >> 30);
              const int val = (int)(sv & 0x3fffffffu) - 1;
              if (kind == 1 && cfirst - val >= need)
                kind = 5;
              const uint32_t stop = group8_bits(wantm && kind != 1);
              const int first = stop ? __ffs(stop) - 1 : 0;
              const int fkind = __shfl(kind, gbase | first);
              const int fval = __shfl(val, gbase | first);
              if (wantm) {
                if (!stop) {
                  look -= 8;
                  progressed = true;  // (the walk goes on at once)
                } else if (fkind == 2) {
                  lin = fval;
                  lin_known = lin_exact = true;
                } else if (fkind == 3) {
                  lin = l_in;  // as the previous level left it
                  lin_known = lin_exact = true;
                } else if (fkind == 5) {
                  lin = fval - 1;  // stands for "no reset within reach"
                  lin_known = true;
                } else {
                  look -= first;  // undecided: everything nearer is reset-free
                }
              }
            }
            // ---- groups that know L settle; what they leave may settle the next group of the
            // wavefront in the same iteration (registers, no memory) ----
            const unsigned long long lead = 0x0101010101010101ull;
            const unsigned long long before = (1ull << gbase) - 1;
            for (int pass = 0; pass < 8; pass++) {
              const bool settle = stage == 1 && zs == 1 && lin_known;
              if (__any(settle)) {
                const uint32_t rr = resets_for(lin, settle);
                if (settle) {
                  zero_r = zeroed(rr, lin);
                  zs = 2;
                  if (rr) {
                    outk = 2;
                    outv = cfirst + (31 - __clz(rr));
                  } else if (lin_exact) {
                    outk = 2;   // the state passes through unchanged, and it is known
                    outv = lin;
                  } else {
                    outk = 1;
                  }
                }
              }
              const bool w2 = stage == 1 && zs == 1 && !lin_known;
              if (!__any(w2))
                break;
              const unsigned long long nt = __ballot(outk != 1) & lead & before;
              const int pl = nt ? 63 - __clzll((long long)nt) : 0;
              const int pk = __shfl(outk, pl);
              const int pv = __shfl(outv, pl);
              const bool found = w2 && nt != 0 && pk == 2;
              if (found) {
                lin = pv;
                lin_known = lin_exact = true;
              }
              if (!__any(found))
                break;
            }
          }
          can = stage == 1 && zs == 2;
          if (can && t == 0) {
            // what this block does to L: its last reset (or the known state passing through), or
            // nothing (then the word carries the block's first coefficient index for the walk)
            rs[p] = outk >= 2 ? sweep_word(2, outv) : sweep_word(1, cfirst);
            if (outk >= 2)
              atomicMax(&l_cur_s, outv);  // L carried to the next level
          }
        }

        prof.mark<2>();
        if (__any(can)) {
          progressed = true;
          // ---- (W) coefficients, DC, inverse transform, commit ---------------
          VT pw_[C];
#pragma unroll
          for (int k = 0; k < C; k++)
            pw_[k] = pt[k];
          bool zero_me = false;
          if (kLossy)
            zero_me = __shfl((int)zero_r, gbase | (crank & 7)) != 0 && coded;
          if (coded && can) {
#pragma unroll
            for (int k = 0; k < C; k++) {
              int32_t co;
              if (kEnc) {
                co = zero_me ? 0 : qc[k];
                cplane[(size_t)k * n_s] = co;
              } else {
                co = qc[k];
              }
              pw_[k] += A::dequant_fp(qaa[k ? 1 : 0], co);
            }
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
            const bool both = (bothm >> st) & 1u;
            const bool swap = (swapm >> st) & 1u;
#pragma unroll
            for (int k = 0; k < C; k++) {
              const VT own = pw_[k], oth = shfl_xor_v(own, bit);
              if (both) {
                pw_[k] = left ? A::mulc(own, ca[st]) - A::mulc(oth, cb[st])
                              : A::mulc(oth, cb[st]) + A::mulc(own, ca[st]);
              } else if (swap) {
                pw_[k] = oth;
              }
            }
          }
          if (can && has) {
            // the ring first (value, then flag): it is what the next hop of the chain waits for
            VT vn[C];
#pragma unroll
            for (int k = 0; k < C; k++) {
              VT v = pw_[k];
              if (w > 1)
                v = A::mulc(A::shr(v, nrm_shift), nrm_rs);
              v = ext ? v : A::round_int(v);
              vn[k] = v;
              ring_val[(rslot * 64 + lane) * C + k] = __builtin_bit_cast(uint64_t, v);
            }
            ring_flag[rslot * 64 + lane] = my_flag;
#pragma unroll
            for (int k = 0; k < C; k++) {
              const uint64_t vb = __builtin_bit_cast(uint64_t, vn[k]);
              const u32x4 gr = {(uint32_t)vb, (uint32_t)(vb >> 32), mtag, 0u};
              __builtin_amdgcn_raw_buffer_store_b128(gr, mrsrc, (int)((crow * C + k) * 16), 0, /*sc1*/ 16);
            }
#pragma unroll
            for (int k = 0; k < C; k++) {
              par2(ctx.rec_us, cur_par)[crow * C + k] = A::to_i64(ext ? pw_[k] : A::round_int(A::muli(pw_[k], 4)));
              par2(ctx.rec, cur_par)[crow * C + k] = A::to_i64(vn[k]);
            }
            par2(ctx.nneigh, cur_par)[crow] = inherit_dc ? neigh_count : 19;
          }
          if (can)
            stage = 3;
        }

        prof.mark<3>();
        if (!progressed) {
          prof.idle();
          ++spins;
          if (spins > (1u << 22) && lane == 0)
            atomicExch(ctx.error, 1);  // fail loudly instead of hanging the GPU
          if ((spins & 4095u) == 0 && __hip_atomic_load(ctx.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            break;  // (wave-uniform: spins is)
          __builtin_amdgcn_s_sleep(1);
        }
      }
      prof.round_end(lane, li);
    }
    // the level is complete: what it stored is the next level's input (ordinary loads)
    __threadfence();
    __syncthreads();
    __threadfence();
  }
  // the zero-run state for the level kernels that go on below this launch: they read the entry of the
  // OTHER parity of their level (the prepass copies it over, raht_levels.hpp)
  if (kLossy && threadIdx.x == 0)
    ctx.slice_l[(sw.li_lo & 1) * tv.num_slices + s] = l_cur_s;
  // ArithF64: a value left the range in which doubles are exact -- the sticky word stops every
  // later kernel of the call and the call is redone with ArithI64 (host tier) or reports
  // GPCC_ERR_RANGE (device tier)
  if (A::kF64 && __any(!in_range) && lane == 0)
    atomicCAS(ctx.error, 0, 3);
}


// ---- host side: record layout and the three launches -----------------------------------------
// rbase of the levels [li_lo, li_hi] from the per-level node counts (TreeStats::nodes); returns the
// number of parents the records must hold
inline int64_t
sweep_rec_layout(SweepRec* rec, const int32_t* nodes, int li_hi, int li_lo)
{
  int64_t total = 0;
  for (int li = 0; li < kMaxLevels; li++)
    rec->rbase[li] = 0;
  for (int li = li_hi; li >= li_lo; li--) {
    rec->rbase[li] = (int32_t)total;
    total += nodes[li + 1];
  }
  return total;
}

#ifndef GPCC_SWEEP_NT
#define GPCC_SWEEP_NT 512  // threads of the walking workgroup (8 wavefronts; experiments: 64 .. 512)
#endif
// occupancies, records, the walk: `rec` carved (sweep_rec_carve) for sweep_rec_layout's parents
template<int C>
inline void
sweep_launch(
  hipStream_t st, const LevelCtx& lc, const SweepCtx& sw, const SweepRec& rec, const int32_t* nodes, int num_slices,
  bool encoder, bool f64)
{
  int64_t maxp = 1;
  for (int li = sw.li_hi; li >= sw.li_lo; li--)
    maxp = nodes[li + 1] > maxp ? nodes[li + 1] : maxp;
  const int nlv = sw.li_hi - sw.li_lo + 1;
  const int ogrid = (int)((maxp + 255) / 256 < 1024 ? (maxp + 255) / 256 : 1024);
  const int rgrid = (int)((maxp + 31) / 32 < 4096 ? (maxp + 31) / 32 : 4096);
  hipLaunchKernelGGL(sweep_occ_kernel, dim3(ogrid, nlv), dim3(256), 0, st, lc.tv, sw, rec);
  if (!encoder) {
    if (f64) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sweep_record_kernel<C, false, ArithF64>), dim3(rgrid, nlv), dim3(256), 0, st, lc, sw, rec);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sub_sweep_kernel<C, kSynth, ArithF64, GPCC_SWEEP_NT>), dim3(num_slices), dim3(GPCC_SWEEP_NT), 0, st, lc, sw, rec);
    } else {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sweep_record_kernel<C, false, ArithI64>), dim3(rgrid, nlv), dim3(256), 0, st, lc, sw, rec);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sub_sweep_kernel<C, kSynth, ArithI64, GPCC_SWEEP_NT>), dim3(num_slices), dim3(GPCC_SWEEP_NT), 0, st, lc, sw, rec);
    }
  } else {
    if (f64) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sweep_record_kernel<C, true, ArithF64>), dim3(rgrid, nlv), dim3(256), 0, st, lc, sw, rec);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sub_sweep_kernel<C, kLossySub, ArithF64, GPCC_SWEEP_NT>), dim3(num_slices), dim3(GPCC_SWEEP_NT), 0, st, lc, sw, rec);
    } else {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sweep_record_kernel<C, true, ArithI64>), dim3(rgrid, nlv), dim3(256), 0, st, lc, sw, rec);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sub_sweep_kernel<C, kLossySub, ArithI64, GPCC_SWEEP_NT>), dim3(num_slices), dim3(GPCC_SWEEP_NT), 0, st, lc, sw, rec);
    }
  }
}

// block records of ONE level by worklist index, for raht_level_sub_kernel<.., REC = true>: behind the level's prepass
// (worklist, work_count[li], pocc are the prepass's), `mem` holds sweep_rec_bytes(max_blocks, C)
template<int C>
inline SweepRec
level_record_launch(
  hipStream_t st, const LevelCtx& lc, int li, void* mem, int64_t max_blocks, bool encoder, bool f64)
{
  SweepRec rec{};
  sweep_rec_carve(&rec, mem, max_blocks, C);
  for (int l = 0; l < kMaxLevels; l++)
    rec.rbase[l] = 0;
  rec.occ = lc.pocc;  // (child occupancy of every parent of the level, by parent index)
  rec.worklist = lc.worklist;
  rec.work_count = lc.work_count;
  const SweepCtx sw{li, li};
  const int rgrid = (int)((max_blocks + 31) / 32 < 4096 ? (max_blocks + 31) / 32 : 4096);
  if (!encoder) {
    if (f64)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sweep_record_kernel<C, false, ArithF64>), dim3(rgrid < 1 ? 1 : rgrid), dim3(256), 0, st, lc, sw, rec);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sweep_record_kernel<C, false, ArithI64>), dim3(rgrid < 1 ? 1 : rgrid), dim3(256), 0, st, lc, sw, rec);
  } else {
    if (f64)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sweep_record_kernel<C, true, ArithF64>), dim3(rgrid < 1 ? 1 : rgrid), dim3(256), 0, st, lc, sw, rec);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_sweep_record_kernel<C, true, ArithI64>), dim3(rgrid < 1 ? 1 : rgrid), dim3(256), 0, st, lc, sw, rec);
  }
  return rec;
}

// blocks a level can hold at most (what the host knows from the node counts): every block has a parent, and with the
// RAHT extension at least two children
inline int64_t
level_max_blocks(const int32_t* nodes, int li, bool ext)
{
  const int64_t parents = nodes[li + 1], by_children = (int64_t)nodes[li] / 2 + 1;
  return ext && by_children < parents ? by_children : parents;
}

}  // namespace gpcc
