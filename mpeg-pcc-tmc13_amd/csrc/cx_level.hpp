// cx_level.hpp -- the COMPACT level pass of RAHT without sub-node prediction
// (tmc3/RAHT.cpp:1306-1808 with raht_subnode_prediction_enabled_flag = 0 and the
// RAHT extension): ONE LANE PER CHILD of a branching block, one launch per level.
//
// The tile pass (raht_tile.hpp) visits every parent of a level -- three of four
// are single-child links of a chain in a lidar frame -- and gives every block
// eight lanes, of which two or three hold a child.  Here
//   * only blocks with >= 2 children exist (lists of cx_tree.hpp); a chain of
//     single-child nodes is one value slot, never copied level by level;
//   * a wavefront takes ~56 consecutive real children (whole blocks), LANE =
//     CHILD: every lane searches the 6 neighbours of ITS octant (3 faces, 3 edges;
//     the parent is the seventh), sums its own prediction -- no 19-step broadcast --
//     and the butterflies pair lanes through ds_bpermute with the partner derived
//     from the block's occupancy;
//   * the lossy encoder resolves the RDOQ zero-run state (tmc3/RAHT.cpp:1576-1670)
//     inside the pass: a wavefront's coefficients are one 64-wide rdoq_chunk, the
//     state is handed from wavefront to wavefront by decoupled look-back (words
//     as in raht_rdoq.hpp) -- one launch per level instead of analyze / resolve /
//     synthesis, and no transformed prediction written to memory and read back.
// Wavefronts do not synchronise with each other (no barrier after the table load).
#pragma once

#include "cx_tree.hpp"
#include "raht_levels.hpp"
#include "raht_rdoq.hpp"

namespace gpcc {

constexpr int kCxG = 57;        // ranks per wavefront; whole blocks: <= 57 + 7 lanes
constexpr int kCxSlices = 64;   // slices whose plan is staged in LDS (more: read from memory)

struct CxCtx {
  TreeView tv;
  CxLists cl;
  const gpcc_raht_params* params;
  const SliceSched* sched;
  const int32_t* attr_prefix;  // P[N+1][C] (encoder)
  int64_t* val;                // [2N][C] unscaled reconstruction of every value slot
  int64_t* rec;                // [2N][C] scaled reconstruction
  int32_t* nn;                 // [2N]    numParentNeigh of the node that was reconstructed there
  int32_t* coeffs;             // planar per slice
  const SharedLut* lut;
  unsigned long long* tstate;  // [tiles] RDOQ look-back words of this level
  int32_t* slice_l;            // [2][S] last RDOQ reset, by the level's parity in the slice's plan
  int32_t li;                  // children level of this launch
};

struct CxSlice {
  int32_t sp0, sp1;   // the slice's nodes in level li + 1
  int32_t sc0;        // first node of the slice in level li
  int32_t pt0, n_s;
  int32_t coeff_end;  // one past the level's last coefficient (slice relative)
  uint32_t procmask;  // bit l: level l is processed in this slice
  LevelSched e;
};

struct CxSmem {
  SharedLut lut;
  int32_t pw[19];
  uint8_t noff[20];
  uint8_t nid[8][8];  // the 6 neighbours of every octant
  int32_t nsl;
  CxSlice sl[kCxSlices];
  uint32_t occ[4][32];
  uint32_t found[4][32];
  uint32_t desc[4][64];
};

__device__ __forceinline__ CxSlice
cx_load_slice(const CxCtx& cx, int li, int s)
{
  const TreeView& tv = cx.tv;
  CxSlice r;
  r.sp0 = tv.soff[li + 1][s];
  r.sp1 = tv.soff[li + 1][s + 1];
  r.sc0 = tv.soff[li][s];
  r.pt0 = tv.pt_off[s];
  r.n_s = tv.pt_off[s + 1] - r.pt0;
  const SliceSched* sc = &cx.sched[s];
  r.e = sc->lvl[li];
  const int m = tv.soff[li][s + 1] - r.sc0;
  r.coeff_end = r.e.coeff_base + (r.e.is_root ? m : m - (r.sp1 - r.sp0));
  uint32_t pm = 0;
  for (int l = 0; l < tv.nlev; l++)
    pm |= (uint32_t)(sc->lvl[l].processed != 0) << l;
  r.procmask = pm;
  return r;
}

// slice of parent j (level li + 1) and its plan
__device__ __forceinline__ CxSlice
cx_slice_of(const CxCtx& cx, const CxSmem& sm, int li, int j, int* s_out)
{
  const int S = cx.tv.num_slices;
  if (S <= kCxSlices) {
    int lo = 0, hi = S - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sm.sl[mid].sp1 <= j)
        lo = mid + 1;
      else
        hi = mid;
    }
    *s_out = lo;
    return sm.sl[lo];
  }
  const int s = find_slice(cx.tv.soff[li + 1], S, j);
  *s_out = s;
  return cx_load_slice(cx, li, s);
}

// 8 presence bits -> a mask of 8 nibbles
__device__ __forceinline__ uint32_t
cx_nibbles(uint32_t x)
{
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x * 15u;
}

__device__ __forceinline__ int64_t
cx_bperm_i64(int src_lane, int64_t v)
{
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(v >> 32));
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

template<int C, bool ENC>
__global__ __launch_bounds__(256) void
cx_level_kernel(CxCtx cx)
{
  __shared__ CxSmem sm;
  const TreeView& tv = cx.tv;
  if (tree_failed(tv))
    return;
  const int li = cx.li;
  const int L = li + 1;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const ParamsConst prm = (ParamsConst)cx.params;
  const int S = tv.num_slices;
  const int n = tv.n_total;

  // ---- tables shared by the workgroup ---------------------------------------------
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(cx.lut);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&sm.lut);
    for (int i = threadIdx.x; i < (int)(sizeof(SharedLut) / 4); i += blockDim.x)
      dst[i] = src[i];
    if (threadIdx.x < 19) {
      sm.pw[threadIdx.x] = prm->pred_weight_parent[threadIdx.x];
      sm.noff[threadIdx.x] = (uint8_t)neigh_offset(threadIdx.x);
    }
    if (threadIdx.x >= 64 && threadIdx.x < 72) {
      const int o = threadIdx.x - 64;
      int cnt = 0;
      for (int i = 1; i < 19; i++)
        if ((neigh_mask(i) >> o) & 1)
          sm.nid[o][cnt++] = (uint8_t)i;
    }
    if (S <= kCxSlices && (int)threadIdx.x >= 128 && (int)threadIdx.x < 128 + S)
      sm.sl[threadIdx.x - 128] = cx_load_slice(cx, li, threadIdx.x - 128);
    __syncthreads();
  }
  const SharedLut& lut = sm.lut;

  const CxLevelTab* __restrict__ tab = cx.cl.tab;
  const int R = tab->nr[li];
  const int nb = tab->nb[li];
  const int ntiles = (R + kCxG - 1) / kCxG;
  const int tile = (int)blockIdx.x * 4 + wave;
  if (tile >= ntiles)
    return;
  const int32_t* __restrict__ bp = cx.cl.bp + tab->boff[li];
  const int32_t* __restrict__ bq = cx.cl.bq + tab->boff[li] + li;
  const int32_t* __restrict__ rb = cx.cl.rb + tab->roff[li];

  // ---- the tile: blocks whose first child has a rank in [tile G, (tile + 1) G) -----
  int b0, b1;
  {
    const int r_lo = tile * kCxG;
    const int r_hi = r_lo + kCxG;
    b0 = rb[r_lo];
    if (bq[b0] < r_lo)
      b0++;
    if (r_hi >= R) {
      b1 = nb;
    } else {
      b1 = rb[r_hi];
      if (bq[b1] < r_hi)
        b1++;
    }
  }
  const int r0 = bq[b0];
  const int nch = bq[b1] - r0;
  if (nch <= 0) {
    // (the last tile of a level when the block before it took its ranks): the
    // zero-run state passes through
    if (ENC && lane == 0)
      __hip_atomic_store(
        &cx.tstate[tile], ((unsigned long long)(li + 1) << 48) | (1ull << 32), __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const bool on = lane < nch;
  uint32_t* wocc = sm.occ[wave];
  uint32_t* wfound = sm.found[wave];
  uint32_t* wdesc = sm.desc[wave];
  if (lane < 32) {
    wocc[lane] = 0;
    wfound[lane] = 0;
  }
  __builtin_amdgcn_wave_barrier();  // (LDS words of the wavefront: written and read in program order)

  // ---- lane = child: its block, its parent, itself ----------------------------------
  const int r = r0 + (on ? lane : 0);
  const int b = rb[r];
  const int bl = b - b0;                 // block of the tile
  const int j = bp[b];                   // parent (level li + 1)
  const int q0 = bq[b];
  const int u = r - q0;                  // ordinal among the block's children
  const int base_lane = q0 - r0;         // lane of the block's first child
  const int c0 = tv.fc[L][j];
  const int64_t pkey = tv.key[L][j];
  const uint32_t phold = cx.cl.hold[L][j];
  int s;
  const CxSlice sl = cx_slice_of(cx, sm, li, j, &s);
  const LevelSched e = sl.e;
  const int child = c0 + u;
  const int fa = tv.fp[li][child], fb = tv.fp[li][child + 1];
  const int oct = (int)(tv.key[li][child] & 7);
  const int32_t w = fb - fa;
  const int cslot = u ? fa : n + fb;
  const int pslot = (int)(phold & kCxSlotMask);
  const int ptop = (int)(phold >> kCxSlotBits);
  const bool inherit_dc = !e.is_root;

  atomicOr(&wocc[bl], on ? 1u << oct : 0u);
#ifdef CX_DEBUG
  if (lane < 4 || !(fa >= 0 && fb <= n)) fprintf(stderr, "li %d tile %d lane %d on %d r %d b %d j %d q0 %d u %d c0 %d child %d fa %d fb %d b0 %d b1 %d r0 %d nch %d R %d nb %d\n", li, tile, lane, (int)on, r, b, j, q0, u, c0, child, fa, fb, b0, b1, r0, nch, R, nb);
#endif

  // source sum of the child: difference of the modular prefix sums (the reference
  // accumulates these sums in `int` as well, tmc3/RAHT.cpp:131,196)
  int64_t src[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    src[k] = 0;
  if (ENC) {
#pragma unroll
    for (int k = 0; k < C; k++)
      src[k] = fp_from_int((int32_t)(
        (uint32_t)cx.attr_prefix[(size_t)fb * C + k] - (uint32_t)cx.attr_prefix[(size_t)fa * C + k]));
  }

  // the parent's values.  numParentNeigh: 19 when the parent came down a chain
  // through a level its slice processes (the single-child copy sets it,
  // tmc3/RAHT.cpp:1382-1401), else what its own block left
  int64_t pval[C], prec[C];
  int pneigh = 0;
#pragma unroll
  for (int k = 0; k < C; k++)
    pval[k] = prec[k] = 0;
  if (inherit_dc) {
#pragma unroll
    for (int k = 0; k < C; k++) {
      pval[k] = cx.val[(size_t)pslot * C + k];
      prec[k] = cx.rec[(size_t)pslot * C + k];
    }
    const uint32_t through = ptop > L ? (sl.procmask >> L) & ((1u << (ptop - L)) - 1u) : 0u;
    pneigh = through ? 19 : cx.nn[pslot];
  }

  __builtin_amdgcn_wave_barrier();
  const uint32_t occ = wocc[bl];

  // ---- inter-level prediction gating (tmc3/RAHT.cpp:1391-1432) ----------------------
  const bool pred_in_level = on && inherit_dc && prm->raht_prediction_enabled_flag != 0;
  const bool do_search = pred_in_level && pneigh >= prm->raht_prediction_threshold0;

  // ---- neighbour search: the parents at the 6 positions this octant predicts from
  //      (findNeighbours, tmc3/RAHT.cpp:299-368; findNeighbour :272-293 is a
  //      lower_bound limited to raht_prediction_search_range entries either side) -----
  int nq[6];
  {
    const int64_t* __restrict__ pk = tv.key[L];
    const uint64_t mbase = morton3d_add((uint64_t)pkey, ~0ull);
    const int64_t range = prm->raht_prediction_search_range;
    int lo[6], hi[6], end[6];
    int64_t want[6];
#pragma unroll
    for (int t = 0; t < 6; t++) {
      const int id = sm.nid[oct][t];
      const int64_t np = (int64_t)morton3d_add(mbase, sm.noff[id]);
      int64_t d = np - pkey;
      int ga, gb;
      if (d >= 0) {
        d = d >= range ? range : d;
        ga = j;
        gb = (d + 1 < (int64_t)(sl.sp1 - j)) ? j + (int)(d + 1) : sl.sp1;
      } else {
        d = (-d) >= range ? range : -d;
        gb = j;
        ga = (d < (int64_t)(j - sl.sp0)) ? j - (int)d : sl.sp0;
      }
      want[t] = np;
      lo[t] = do_search ? ga : 0;
      hi[t] = end[t] = do_search ? gb : 0;
    }
    while (__any((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2]) | (lo[3] < hi[3])
                 | (lo[4] < hi[4]) | (lo[5] < hi[5]))) {
      int mid[6];
      int64_t kv[6];
#pragma unroll
      for (int t = 0; t < 6; t++) {
        mid[t] = lo[t] + ((hi[t] - lo[t]) >> 1);
        kv[t] = lo[t] < hi[t] ? pk[mid[t]] : 0;
      }
#pragma unroll
      for (int t = 0; t < 6; t++) {
        if (lo[t] < hi[t]) {
          if (kv[t] < want[t])
            lo[t] = mid[t] + 1;
          else
            hi[t] = mid[t];
        }
      }
    }
    int64_t kf[6];
#pragma unroll
    for (int t = 0; t < 6; t++)
      kf[t] = lo[t] < end[t] ? pk[lo[t]] : -1;
    uint32_t fmask = 0;
#pragma unroll
    for (int t = 0; t < 6; t++) {
      const bool hit = lo[t] < end[t] && kf[t] == want[t];
      nq[t] = hit ? lo[t] : -1;
      if (hit)
        fmask |= 1u << sm.nid[oct][t];
    }
    atomicOr(&wfound[bl], fmask);
    __builtin_amdgcn_wave_barrier();
  }
  int neigh_count = 0;
  bool enable_pred = false;
  if (do_search) {
    neigh_count = popc32(wfound[bl]) + 1;
    enable_pred = neigh_count >= prm->raht_prediction_threshold1;
  }

  // ---- intraDcPred for this child (tmc3/RAHT.cpp:421-589, parent-level part) ---------
  int64_t pred[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    pred[k] = 0;
  {
    const bool run = enable_pred;
    uint32_t nh[6];
#pragma unroll
    for (int t = 0; t < 6; t++)
      nh[t] = (run && nq[t] >= 0) ? cx.cl.hold[L][nq[t]] : 0u;
    int64_t nv[6][C];
#pragma unroll
    for (int t = 0; t < 6; t++)
#pragma unroll
      for (int k = 0; k < C; k++)
        nv[t][k] = (run && nq[t] >= 0) ? cx.rec[(size_t)(nh[t] & kCxSlotMask) * C + k] : 0;
    if (run) {
      const int64_t lim_lo = 2 * prec[0], lim_hi = 25 * prec[0];
      int wsum = sm.pw[0];
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] = prec[k] * (int64_t)sm.pw[0];
#pragma unroll
      for (int t = 0; t < 6; t++) {
        if (nq[t] >= 0 && !(10 * nv[t][0] <= lim_lo || 10 * nv[t][0] >= lim_hi)) {
          const int64_t pwt = sm.pw[sm.nid[oct][t]];
          wsum += (int)pwt;
#pragma unroll
          for (int k = 0; k < C; k++)
            pred[k] += nv[t][k] * pwt;
        }
      }
      const int64_t div = pred_divisor(wsum);
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] = fp_mul_c(pred[k], div);
    }
  }

  // ---- normalise (tmc3/RAHT.cpp:1445-1499) -------------------------------------------
  if (on && w > 1) {
    if (ENC) {
#pragma unroll
      for (int k = 0; k < C; k++)
        src[k] = scale_rsqrt(src[k], w, lut);
    }
    if (enable_pred) {
      const int64_t sq = sqrt_weight(w, lut);
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] = fp_mul_c(pred[k], sq);
    }
  }

  // ---- forward butterflies (fwdTransformBlock222, tmc3/RAHT.cpp:671-701; weights as
  //      mkWeightTree :742).  A value stays in its lane; `pos` is where it sits in
  //      the 2x2x2 block, P the occupied positions, M the lane (nibble) of every
  //      position.  Stage s pairs pos and pos ^ (1 << s): both present -> butterfly
  //      (low-pass left, high-pass right); only the right one -> it moves left. -------
  uint32_t P = occ;
  uint32_t M = 0;
  {
    // the u-th child of the block sits at the u-th occupied octant
#pragma unroll
    for (int p = 0; p < 8; p++)
      M |= (uint32_t)popc32(occ & ((1u << p) - 1u)) << (4 * p);
  }
  int pos = oct;
  int32_t cw = on ? w : 0;
  bool st_both[3], st_left[3];
  int st_lane[3];
  int32_t st_a[3], st_b[3];
#pragma unroll
  for (int st = 0; st < 3; st++) {
    const int bit = 1 << st;
    const uint32_t lm = st == 0 ? 0x55u : (st == 1 ? 0x33u : 0x0Fu);
    const int pp = pos ^ bit;
    const bool partner = on && ((P >> pp) & 1);
    const bool left = !(pos & bit);
    const int plane = base_lane + (int)((M >> (4 * pp)) & 15u);
    const int from = partner ? plane : lane;
    const int32_t ow = __builtin_amdgcn_ds_bpermute(from << 2, cw);
    const int32_t wl = left ? cw : ow, wr = left ? ow : cw;
    int64_t ca = 0, cb = 0;
    if (partner)
      raht_coeffs(wl, wr, lut, &ca, &cb);
    st_both[st] = partner;
    st_left[st] = left;
    st_lane[st] = from;
    st_a[st] = (int32_t)ca;
    st_b[st] = (int32_t)cb;
#pragma unroll
    for (int k = 0; k < C; k++) {
      if (ENC) {
        const int64_t own = src[k], oth = cx_bperm_i64(from, own);
        if (partner)
          src[k] = left ? fp_mul_c(oth, cb) + fp_mul_c(own, ca) : fp_mul_c(own, ca) - fp_mul_c(oth, cb);
      }
      {
        const int64_t own = pred[k], oth = cx_bperm_i64(from, own);
        // (a block's lanes share enable_pred)
        if (partner && enable_pred)
          pred[k] = left ? fp_mul_c(oth, cb) + fp_mul_c(own, ca) : fp_mul_c(own, ca) - fp_mul_c(oth, cb);
      }
    }
    if (partner)
      cw = wl + wr;
    // positions after the stage
    const uint32_t Pl = P & lm, Pr = (P >> bit) & lm;
    const uint32_t moved = cx_nibbles(Pr & ~Pl);
    M = (M & ~moved) | ((M >> (4 * bit)) & moved);
    if (!left && !partner)
      pos = pp;
    P = (Pl | Pr) | ((Pl & Pr) << bit);
  }

  // ---- coefficient slot of this lane's position (scanBlock :776-791) -------------------
  const uint32_t pscan = ((P >> 0) & 1) | (((P >> 4) & 1) << 1) | (((P >> 2) & 1) << 2)
    | (((P >> 1) & 1) << 3) | (((P >> 6) & 1) << 4) | (((P >> 5) & 1) << 5)
    | (((P >> 3) & 1) << 6) | (((P >> 7) & 1) << 7);
  const int spos = (0x74516230u >> (4 * pos)) & 7;  // scan order 0,4,2,1,6,5,3,7
  const int rank = popc32(pscan & ((1u << spos) - 1u));
  const bool coded = on && (pos != 0 || !inherit_dc);
  const int pj = j - sl.sp0;
  const int cidx = e.coeff_base + (inherit_dc ? (c0 - sl.sc0) - pj + rank - 1 : rank);
  int32_t* __restrict__ cplane = cx.coeffs + (size_t)sl.pt0 * C + cidx;
  const int cblock = cidx - (inherit_dc ? rank - 1 : rank);  // the block's first coefficient

  Quantizer qa[2];
  {
    int ac0 = 0, ac1 = 0;
    if (e.ac_layer < prm->num_ac_qp_layers && pos) {
      ac0 = prm->ac_qp_offset[e.ac_layer][pos - 1][0];
      ac1 = prm->ac_qp_offset[e.ac_layer][pos - 1][1];
    }
    qpset_quantizers(prm, e.qp_layer, ac0, ac1, qa);
  }

  int32_t qco[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    qco[k] = 0;
  if (ENC) {
    // ---- residual, RDOQ statistics (tmc3/RAHT.cpp:1584-1616), quantisation -----------
    uint32_t d = kDescZero;
    if (coded) {
      if (enable_pred) {
#pragma unroll
        for (int k = 0; k < C; k++)
          src[k] -= pred[k];
      }
      Quantizer qr[2];
      qpset_quantizers(prm, e.qp_layer, 0, 0, qr);
      int64_t sum_coeff = 0, dist2 = 0;
      int rate_coeff = 0;
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int64_t co = fp_round(src[k]);
        dist2 += co * co;
        int64_t aq = quantize(qr[k ? 1 : 0], co * 256);
        aq = aq < 0 ? -aq : aq;
        sum_coeff += aq;
        rate_coeff += rate_log_small(aq);
        qco[k] = (int32_t)quantize(qa[k ? 1 : 0], co * 256);
      }
      d = kDescNever;
      if (sum_coeff < 3) {
        const int64_t l0 = qr[0].step;
        const int64_t lambda = l0 * l0 * (C == 1 ? 25 : 35);
        d = rdoq_threshold(dist2, lambda, rate_coeff, (uint32_t)sl.n_s);
        if (sum_coeff == 0)
          d |= kDescZero;
      }
    }

    // ---- the zero-run state (tmc3/RAHT.cpp:1618-1669): the tile's coefficients in
    //      coding order are one chunk per slice; the incoming last reset comes from
    //      the wavefront before (look-back) or from the level before (slice_l) -----------
    const unsigned long long ep = (unsigned long long)(li + 1) << 48;
    unsigned long long todo = __ballot(coded);
    bool zero_me = false;
    bool published = false;
    while (todo) {
      const int first = __ffsll((long long)todo) - 1;
      const int s_cur = __shfl(s, first);
      // (lanes are in child order, coefficients in scan order: the segment's first
      // coefficient is the first of its first block)
      const int c_first = __shfl(cblock, first);
      const bool mine = coded && s == s_cur;
      const unsigned long long seg = __ballot(mine);
      todo &= ~seg;
      const bool last_seg = todo == 0;
      const int ncoef = __popcll(seg);
      const int e_base = __shfl(e.coeff_base, first);
      const int e_par = __shfl((int)e.parity, first);
      const int c_end = __shfl(sl.coeff_end, first);
      if (mine)
        wdesc[cidx - c_first] = d;
      __builtin_amdgcn_wave_barrier();
      const bool valid = lane < ncoef;
      const uint32_t dd = valid ? wdesc[lane] : kDescZero;
      const int ci = c_first + lane;
      const bool starts = c_first == e_base;      // the slice's first coefficient of the level
      const bool ends = c_first + ncoef == c_end;  // ... and its last
      int l_in = 0, l_out = 0;
      int tz;
      if (starts) {
        l_in = cx.slice_l[(e_par ^ 1) * S + s_cur];
        l_out = rdoq_chunk(dd, ci, valid, l_in, c_first, &tz);
        if (last_seg && lane == 0)
          __hip_atomic_store(
            &cx.tstate[tile], ep | (3ull << 32) | (uint32_t)l_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = published || last_seg;
      } else {
        // (only the tile's first segment continues a slice from the tile before)
        const int la0 = -1, lb0 = c_first - 1;
        const int la = rdoq_chunk(dd, ci, valid, la0, c_first, &tz);
        const int lb = rdoq_chunk(dd, ci, valid, lb0, c_first, &tz);
        const int status = lb == lb0 ? kTileTransparent : (la == lb ? kTileClosed : kTileOpen);
        if (last_seg && status != kTileOpen) {
          if (lane == 0)
            __hip_atomic_store(
              &cx.tstate[tile],
              ep | ((unsigned long long)(status == kTileClosed ? 2 : 1) << 32) | (uint32_t)la,
              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          published = true;
        }
        bool have = false;
        int k0 = tile - 1;  // lane v looks at tile k0 - v
        unsigned spins = 0;
        while (!have) {
          const int kt = k0 - lane;
          const bool in = kt >= 0;
          unsigned long long wv = 0;
          if (in)
            wv = __hip_atomic_load(&cx.tstate[kt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool ready = in && (wv >> 48) == (unsigned long long)(li + 1);
          const int kind = ready ? (int)((wv >> 32) & 0xffff) : 0;
          const unsigned long long decides = __ballot(ready && kind >= 2);
          const unsigned long long notready = __ballot(in && !ready);
          const unsigned long long stop = decides | notready;
          if (!stop) {
            k0 -= kWave;  // 64 transparent tiles: further back (the slice's first tile decides)
            if (k0 < 0) {
              // cannot happen: the walk ends at the tile that starts the slice's level
              if (lane == 0)
                atomicExch(tv.error, 1);
              return;
            }
            continue;
          }
          const int firstw = __ffsll((long long)stop) - 1;
          if ((decides >> firstw) & 1) {
            l_in = (int)(uint32_t)__shfl((int)(uint32_t)wv, firstw);
            have = true;
          } else {
            k0 -= firstw;
            if (++spins > (1u << 22)) {
              if (lane == 0)
                atomicExch(tv.error, 1);
              return;
            }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        l_out = rdoq_chunk(dd, ci, valid, l_in, c_first, &tz);
        if (last_seg && status == kTileOpen) {
          if (lane == 0)
            __hip_atomic_store(
              &cx.tstate[tile], ep | (3ull << 32) | (uint32_t)l_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          published = true;
        }
      }
      if (ends && lane == 0)
        cx.slice_l[e_par * S + s_cur] = l_out;
      // decisions back to the lanes that own the coefficients
      const uint32_t thr = dd & kDescNever;
      const bool zero = valid && thr != kDescNever && (uint32_t)tz >= thr;
      __builtin_amdgcn_wave_barrier();
      wdesc[lane] = zero ? 1u : 0u;
      __builtin_amdgcn_wave_barrier();
      if (mine)
        zero_me = wdesc[cidx - c_first] != 0;
      __builtin_amdgcn_wave_barrier();
    }
    if (!published && lane == 0)  // a tile without a coefficient lets the state pass
      __hip_atomic_store(&cx.tstate[tile], ep | (1ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (coded) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        if (zero_me)
          qco[k] = 0;
        cplane[(size_t)k * sl.n_s] = qco[k];
      }
    }
  } else if (coded) {
#pragma unroll
    for (int k = 0; k < C; k++)
      qco[k] = cplane[(size_t)k * sl.n_s];
  }

  // ---- reconstruction: prediction + de-quantised residual, inherited DC ---------------
  if (!enable_pred) {
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = 0;
  }
  if (coded) {
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] += fp_from_int(dequantize(qa[k ? 1 : 0], qco[k]));
  }
  if (on && inherit_dc && pos == 0) {
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = pval[k];  // (tmc3/RAHT.cpp:1727-1742, extension)
  }

  // ---- inverse butterflies (tmc3/RAHT.cpp:707-737) -------------------------------------
#pragma unroll
  for (int st = 2; st >= 0; st--) {
    const int64_t ca = st_a[st], cb = st_b[st];
#pragma unroll
    for (int k = 0; k < C; k++) {
      const int64_t own = pred[k], oth = cx_bperm_i64(st_lane[st], own);
      if (st_both[st])
        pred[k] = st_left[st] ? fp_mul_c(own, ca) - fp_mul_c(oth, cb) : fp_mul_c(oth, cb) + fp_mul_c(own, ca);
    }
  }

  // ---- the child's values (:1754-1806) ---------------------------------------------------
  if (on) {
#pragma unroll
    for (int k = 0; k < C; k++) {
      int64_t v = pred[k];
      cx.val[(size_t)cslot * C + k] = v;
      if (w > 1)
        v = scale_rsqrt(v, w, lut);
      cx.rec[(size_t)cslot * C + k] = v;
    }
    cx.nn[cslot] = inherit_dc ? neigh_count : 19;
  }
}

}  // namespace gpcc
