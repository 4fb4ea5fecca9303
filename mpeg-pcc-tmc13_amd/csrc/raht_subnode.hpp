// raht_subnode.hpp -- level kernel for raht_subnode_prediction_enabled_flag=1
// (the CTC default): the prediction of a block additionally uses the
// ALREADY RECONSTRUCTED children of up to 12 causal neighbour parents of the
// same level (tmc3/RAHT.cpp:370-415, 503-565).  A neighbour parent counts
// only once it has been processed (its `occupancy` is zeroed at level start
// :1223 and set at :1393), and blocks are processed in Morton order, so
// block j depends on neighbour blocks q < j: a wavefront of dependencies
// sweeps each level (about 2 000 sequential steps per slice).
//
// Execution model.  Wavefronts claim 8 consecutive worklist blocks with a
// ticket (8 counters, workgroup index mod 8, so claiming is monotone per
// counter and the lowest unfinished block can always run: every dependency
// points to a lower block).  Which term a (child lane, neighbour) pair
// contributes is decided by the tree alone -- the neighbour parent's value,
// a child copied by the prepass, or a child reconstructed by THIS launch --
// so everything but the last kind (children gather, butterfly coefficients,
// neighbour search, parent-level terms, weight sum, divisor, the encoder's
// forward transform of the source) is settled before the wait.  A
// reconstructed child is published as ONE 16-byte write-through (sc1) store
// {value, tag} into a mailbox next to the plain store later launches read;
// a wave-uniform loop polls the granules a block still needs, one per lane
// and iteration (cdna_hip_programming.md G16, form R2: the data is the flag
// -- one memory round trip per dependency hop, no flag word, no drain), and
// finishes the groups whose values have all arrived: normalise, transform,
// coefficients, inverse, reconstruction.  Children of a neighbour claimed by
// the SAME wavefront (Morton-adjacent blocks: most hops of the longest
// chains) are taken from its registers instead.  The poll is the only
// vector-memory wait inside the loop -- on gfx9 any s_waitcnt vmcnt(0) also
// waits for the write-through stores in flight.  Spins
// are bounded: a stuck launch raises ctx.error instead of hanging the GPU.
//
// Modes: kSynth (decoder), kFused (integer-Haar encoder) and kLossySub, the
// lossy encoder.  There the RDOQ zero-run state (tmc3/RAHT.cpp:1618-1669)
// is a second dependency: a block's decisions need L, the index of the last
// reset before its first coefficient (raht_rdoq.hpp).  Each block evaluates
// its <= 8 coefficients under the two extreme hypotheses for L; when both
// agree it commits at once and publishes a state word (final L, or
// "reset-free" + its first coefficient index).  Otherwise only a reset within
// the reach of its thresholds matters, so it walks the predecessors' words
// -- registers inside the wavefront, then memory, 8 per step -- until a
// block with a reset, the slice start, or `reach` reset-free coefficients.
#pragma once

#include <stdlib.h>

#include "raht_arith.hpp"
#include "raht_inter.hpp"
#include "raht_levels.hpp"
#include "raht_links.hpp"

namespace gpcc {

// Which round the `tk`-th claim of class `cls` (workgroup index mod 8 -- the XCD, with the round-robin placement
// of workgroups) takes: GPCC_SUB_CHUNK = K consecutive rounds per class and turn (K = 1: rounds interleave,
// round r belongs to class r mod 8).  Monotone in tk for every class.
#ifndef GPCC_SUB_CHUNK
#define GPCC_SUB_CHUNK 1
#endif
#ifndef GPCC_SUB_DSTORE
#define GPCC_SUB_DSTORE 0
#endif
#ifndef GPCC_SUB_POLL_N
#define GPCC_SUB_POLL_N 1
#endif
#ifndef GPCC_SUB_IDLE_FAST
#define GPCC_SUB_IDLE_FAST 1
#endif
#ifndef GPCC_SUB_IDLE_SPIN
#define GPCC_SUB_IDLE_SPIN 0
#endif
__device__ __forceinline__ int64_t
sub_round_of_ticket(int tk, int cls)
{
  constexpr int K = GPCC_SUB_CHUNK;
  return ((int64_t)(tk / K) * 8 + cls) * K + tk % K;
}

// Where a wavefront's time goes (experiment builds only, -DGPCC_SUB_PROF:
// s_memtime around the stages of the loop, summed per launch and per level
// into g_sub_prof, read back with gpcc_debug_sub_prof).  Empty otherwise.
#ifdef GPCC_SUB_PROF
__device__ unsigned long long g_sub_prof[16 + 32 * 20];  // [16 + li * 4 + ..] rounds, prologue, loop, iterations; [144 + li * 6 + ..] stage ticks 0-3, idle iterations
struct SubProf {
  unsigned long long t0 = 0, t1 = 0, last = 0, acc[4] = {0, 0, 0, 0}, iters = 0, idle_iters = 0;
  // finer marks inside (P) and (W): [336 + li * 10 + ..] P executions, W executions, then the sub-steps' cycles
  unsigned long long last2 = 0, sub_acc[6] = {0, 0, 0, 0, 0, 0}, execs[2] = {0, 0};
  template<int WHICH> __device__ void enter() { execs[WHICH]++; last2 = now(); }
  template<int K> __device__ void sub() { const unsigned long long t = now(); sub_acc[K] += t - last2; last2 = t; }
  __device__ static unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
  __device__ void round_begin() { t0 = now(); }
  __device__ void loop_begin() { t1 = now(); }
  __device__ void iter_begin() { last = now(); iters++; }
  template<int STAGE> __device__ void mark() { const unsigned long long t = now(); acc[STAGE] += t - last; last = t; }
  __device__ void idle() { idle_iters++; }
  __device__ void round_end(int lane, int li)
  {
    if (lane != 0)
      return;
    const unsigned long long t2 = now();
    const unsigned long long v[9] = {1, t1 - t0, t2 - t1, acc[0], acc[1], acc[2], acc[3], iters, idle_iters};
    for (int i = 0; i < 9; i++)
      atomicAdd(&g_sub_prof[i], v[i]);
    atomicAdd(&g_sub_prof[16 + li * 4 + 0], 1ull);
    atomicAdd(&g_sub_prof[16 + li * 4 + 1], t1 - t0);
    atomicAdd(&g_sub_prof[16 + li * 4 + 2], t2 - t1);
    atomicAdd(&g_sub_prof[16 + li * 4 + 3], iters);
    for (int i = 0; i < 4; i++)
      atomicAdd(&g_sub_prof[144 + li * 6 + i], acc[i]);
    atomicAdd(&g_sub_prof[144 + li * 6 + 4], idle_iters);
    atomicAdd(&g_sub_prof[336 + li * 10 + 0], execs[0]);
    atomicAdd(&g_sub_prof[336 + li * 10 + 1], execs[1]);
    for (int i = 0; i < 6; i++)
      atomicAdd(&g_sub_prof[336 + li * 10 + 2 + i], sub_acc[i]);
  }
};
#else
struct SubProf {
  __device__ void round_begin() {}
  __device__ void loop_begin() {}
  __device__ void iter_begin() {}
  template<int STAGE> __device__ void mark() {}
  template<int WHICH> __device__ void enter() {}
  template<int K> __device__ void sub() {}
  __device__ void idle() {}
  __device__ void round_end(int, int) {}
};
#endif

// Rounds of 8 blocks a wavefront of the lossy sub-node encoder takes per claim at a level with `parents` parents
// (LevelCtx::claim_rounds; an experiment, 10-20 x slower: profiles/r05_claim_rounds_ab.txt).  One place for both
// drivers (gpcc_attr_mi355.hip launch_transform, raht_inter_driver.hpp inter_run): only in experiment builds,
// GPCC_SUB_CLAIM = R (default 1) for levels with at most GPCC_SUB_CLAIM_PARENTS parents (default 100 000); read once
// per process, except under the emulator, whose tests change the environment between calls.
inline int
sub_claim_rounds(bool encoder, bool haar, int64_t parents)
{
#if GPCC_EXPERIMENTS
  auto rounds = [] {
    const char* e = getenv("GPCC_SUB_CLAIM");
    const int v = e ? atoi(e) : 1;
    return v < 1 ? 1 : (v > 64 ? 64 : v);
  };
  auto limit = [] {
    const char* e = getenv("GPCC_SUB_CLAIM_PARENTS");
    return e ? (int64_t)atoll(e) : (int64_t)100000;
  };
#ifdef GPCC_EMU
  const int r = rounds();
  const int64_t lim = limit();
#else
  static const int r = rounds();
  static const int64_t lim = limit();
#endif
  return (encoder && !haar && r > 1 && parents <= lim) ? r : 1;
#else
  (void)encoder;
  (void)haar;
  (void)parents;
  return 1;
#endif
}

__device__ __forceinline__ int
occu_shift(int i12)
{
  constexpr uint8_t s[12] = {6, 5, 4, 3, 2, 1, 3, 1, 2, 1, 2, 3};
  return s[i12];
}

__device__ __forceinline__ int32_t
load_agent_i32(const int32_t* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int64_t
load_agent_i64(const int64_t* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void
store_agent_i64(int64_t* p, int64_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The lossy encoder needs more live state than 128 registers hold (83 spilled
// VGPRs at 4 waves/SIMD); the kernel is bound by its dependency chains, not
// by occupancy, so it trades a wave for registers: 1M dense C=3 forward
// 46.2 -> 32.7 ms, 1M lidar 15.1 -> 14.6 ms (2 waves: 34.3 / 15.1).
#ifndef GPCC_SUB_SLEEP
#define GPCC_SUB_SLEEP 4
#endif
#ifndef GPCC_SUB_POLL_RR
#define GPCC_SUB_POLL_RR 0
#endif
#ifndef GPCC_SUB_LOSSY_WAVES
#define GPCC_SUB_LOSSY_WAVES 3
#endif
#ifndef GPCC_SUB_SYNTH3_WAVES
#define GPCC_SUB_SYNTH3_WAVES 4
#endif
// A = the arithmetic back end (raht_arith.hpp): ArithI64, or ArithF64 for batches whose values stay
// in the range where doubles are exact (extension on, no integer Haar; the kernel checks).
// INTER: attribute inter prediction (raht_inter.hpp) -- where ctx.inter says so a block that lines up with
// one of the reference frame takes the frame's coefficients as its prediction; such a block waits for nobody.
// (The encoder's second, intra-only candidate of a level is a launch of the kernel WITHOUT the flag on a
// workspace of its own: raht_inter_driver.hpp.)
// REC: the static half of a round's prologue comes from block records by worklist index (ctx.brec, written by
// raht_sweep_record_kernel right behind the level's prepass: raht_sweep.hpp) -- children, weights, butterfly constants,
// normalisers, the source's forward transform, the 18-neighbour search, the neighbours' child tables; a round then
// starts with 19 + 2 C coalesced loads instead of ~25 dependent ones.  OPT-IN (GPCC_REC=1): measured on the MI355X it
// takes 4-7 % off the level kernels and the record pass costs 12-13 % (profiles/r06_rec_ab.txt) -- the prologue is not
// what a level waits for, in a batch either.  Bit-exact (GPU tier with GPCC_REC=1, tests/test_emu_sweep.py).
// Not with region QPs, integer Haar or INTER (the host decides).
template<int C, int MODE, class A = ArithI64, bool INTER = false, bool REC = false>
__global__ __launch_bounds__(256, MODE == kLossySub ? GPCC_SUB_LOSSY_WAVES : (C == 3 ? GPCC_SUB_SYNTH3_WAVES : 4)) void
raht_level_sub_kernel(LevelCtx ctx)
{
  static_assert(!(REC && (INTER || MODE == kFused)), "records: fixed-point intra kernels only");
  static_assert(MODE == kSynth || MODE == kFused || MODE == kLossySub, "mode");
  static_assert(!(A::kF64 && MODE == kFused), "integer Haar is integer arithmetic");
  typedef typename A::T VT;
  typedef typename A::Coef VC;
  constexpr bool kLossy = MODE == kLossySub;
  __shared__ SharedLut lut_s;
  if (tree_failed(ctx.tv))
    return;
  load_lut(&lut_s, ctx.lut);
  const SharedLut& lut = lut_s;

  constexpr bool kEnc = MODE != kSynth;
  constexpr bool kRecon = true;
  const TreeView& tv = ctx.tv;
  const ParamsConst prm = (ParamsConst)ctx.params;
  const int li = ctx.li;
  const int t = threadIdx.x & 7;
  const int lane = lane_id();
  const int gbase = threadIdx.x & 56;  // first lane of this 8-lane group
  const bool haar = !A::kF64 && prm->integer_haar_enable_flag != 0;
  const bool ext = A::kF64 || prm->raht_extension != 0;
  const int32_t epoch = li + 1;
  // the wavefront's mailbox: values of the children committed in the current round
#ifdef GPCC_EMU  // (tests/emu: __shared__ is one static object, eight workgroups run together)
  __shared__ unsigned long long wmail_s[8 * 4 * 64 * C];
  unsigned long long* wm = wmail_s + ((blockIdx.x & 7) * 4 + (threadIdx.x >> 6)) * (64 * C);
#else
  __shared__ unsigned long long wmail_s[4 * 64 * C];
  unsigned long long* wm = wmail_s + (threadIdx.x >> 6) * (64 * C);
#endif
  int pwc12[12];
#pragma unroll
  for (int i12 = 0; i12 < 12; i12++)
    pwc12[i12] = prm->pred_weight_child[i12];
  bool in_range = true;  // (ArithF64: the magnitudes that bound every product, raht_arith.hpp)
  const int cls = blockIdx.x & 7;

  const int num_work = ctx.work_count[li];
  // Claims.  A wavefront takes ONE round of 8 blocks, from one of eight tickets (workgroup index mod 8: rounds
  // of the eight classes interleave).  Round 5 experiment, opt-in (ctx.claim_rounds = R > 1, GPCC_SUB_CLAIM): R
  // CONSECUTIVE rounds from a single ticket, worked through one after the other with the lossy encoder's
  // zero-run state (tmc3/RAHT.cpp:1618-1669) carried from a round's last block to the next round's first in
  // registers, so that the hop between two wavefronts is paid once per 8 R blocks.  Bit-exact, and 10-20 x
  // slower on the MI355X (profiles/r05_claim_rounds_ab.txt): the rounds of a claim start one after the other,
  // each with its ~100 us prologue of dependent loads, and the next claim's blocks wait for all of them -- the
  // one-round claims of many wavefronts overlap exactly that.  Kept for the record and pinned under the emulator.
#if GPCC_EXPERIMENTS
  const int claim_rounds = ctx.claim_rounds > 1 ? ctx.claim_rounds : 1;
  bool stop_all = false;
  for (;;) {
    int tk = 0;
    if (lane == 0)
      tk = atomicAdd(&ctx.ticket[li * 8 + (claim_rounds > 1 ? 0 : cls)], 1);
    tk = __shfl(tk, 0);
    const int64_t wround0 = claim_rounds > 1 ? (int64_t)tk * claim_rounds : sub_round_of_ticket(tk, cls);
    if (wround0 * 8 >= num_work || stop_all)
      break;
    // the zero-run state behind the previous round of this claim, when it is known exactly
    bool carry_known = false;
    int carry_l = -1;
   for (int sr = 0; sr < claim_rounds; sr++) {
    const int64_t wround = wround0 + sr;
    if (wround * 8 >= num_work)
      break;
    SubProf prof;
    prof.round_begin();
    // a bounded wait has expired somewhere: the result is discarded anyway,
    // leave at once instead of spinning through every remaining round
    if (__hip_atomic_load(ctx.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      stop_all = true;
      break;
    }
#else  // the product library: one round per claim, the loop exactly as in round 4
  for (;;) {
    int tk = 0;
    if (lane == 0)
      tk = atomicAdd(&ctx.ticket[li * 8 + cls], 1);
    tk = __shfl(tk, 0);
    const int64_t wround = sub_round_of_ticket(tk, cls);
    if (wround * 8 >= num_work)
      break;
    SubProf prof;
    prof.round_begin();
    // a bounded wait has expired somewhere: the result is discarded anyway,
    // leave at once instead of spinning through every remaining round
    if (__hip_atomic_load(ctx.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      break;
#endif
    const int wi = (int)(wround * 8) + (lane >> 3);
    const bool live = wi < num_work;
    const int j = live ? ctx.worklist[wi] : 0;
    // ---- locate the block -------------------------------------------
    int s = 0;
    LevelSched e;
    e.processed = 0;
    // (REC) this lane's record: field f at rf[f * rln]
    const size_t rln = REC ? (size_t)ctx.brec.lanes : 0;
    const size_t rat = (size_t)wround * 64 + lane;
    const int32_t* __restrict__ rf = REC ? ctx.brec.f32 + rat : nullptr;
    if (live) {
      if constexpr (REC)
        s = rf[kSfSlice * rln];
      else
        s = find_slice(tv.soff[li + 1], tv.num_slices, j);
      e = ctx.sched[s].lvl[li];
    }
    // all shuffles below run in wave-uniform control flow; lanes of dead
    // groups carry zeros and store nothing
    const bool on = live && e.processed;
    const int sp0 = on ? tv.soff[li + 1][s] : 0;      // slice's parents
    const int sp1 = on ? tv.soff[li + 1][s + 1] : 0;
    const int sc0 = on ? tv.soff[li][s] : 0;          // slice's children
    const int pt0 = on ? tv.pt_off[s] : 0;
    const int n_s = on ? tv.pt_off[s + 1] - pt0 : 0;
    uint32_t rpk = 0, rpk2 = 0;
    if constexpr (REC) {
      rpk = (uint32_t)rf[kSfPk * rln];
      rpk2 = on ? (uint32_t)rf[kSfPk2 * rln] : 0u;
    }
    const int c0 = on ? (REC ? rf[kSfC0 * rln] : tv.fc[li + 1][j]) : 0;
    const int nchild = on ? (REC ? popc32(rpk2 & 0xffu) : tv.fc[li + 1][j + 1] - c0) : 0;
    const int pj = j - sp0;
    const int par_par = e.parity ^ 1, cur_par = e.parity;
    const int64_t prow = (int64_t)pt0 + pj;  // parent row in rec buffers

    // ---- children -> positions ---------------------------------------
    uint32_t occ_;
    if constexpr (REC) {
      occ_ = rpk2 & 0xffu;
    } else {
      const int64_t ckey = t < nchild ? tv.key[li][c0 + t] : 0;
      occ_ = group8_or(t < nchild ? 1u << (int)(ckey & 7) : 0u);
    }
    const uint32_t occ = occ_;
    const bool has = (occ >> t) & 1;
    const int child = c0 + popc32(occ & ((1u << t) - 1));
    const int64_t crow = (int64_t)pt0 + (child - sc0);
    int32_t w = 0;
    VT src[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      src[k] = A::zero();
    if constexpr (REC) {
      w = has ? rf[kSfW * rln] : 0;
      if (kEnc) {
#pragma unroll
        for (int k = 0; k < C; k++)
          src[k] = __builtin_bit_cast(VT, ctx.brec.src[(size_t)k * rln + rat]);  // (already transformed)
      }
    } else if (has) {
      const int f0 = tv.fp[li][child], f1 = tv.fp[li][child + 1];
      w = f1 - f0;
      if (kEnc) {
        if (haar) {
          const int32_t* lf = ctx.haar_lf[li];
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] = A::from_int(lf[(size_t)child * C + k]);
        } else {
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] = A::from_int((int32_t)(
              (uint32_t)ctx.attr_prefix[(size_t)f1 * C + k]
              - (uint32_t)ctx.attr_prefix[(size_t)f0 * C + k]));
        }
      }
    }

    // ---- node qp on the way down (see oracle/raht_oracle.c,
    //      descend_block_qp; tmc3/RAHT.cpp:185-189 vs :246-253) ----------
    int32_t nq0 = 0, nq1 = 0;
    if (ctx.asc_qp) {
      int32_t a0 = 0, a1 = 0;
      if (has) {
        a0 = ctx.asc_qp[li][(size_t)child * 2];
        a1 = ctx.asc_qp[li][(size_t)child * 2 + 1];
      }
      // ascent averages of the pair / quad this position belongs to
      int32_t wa = w, b0 = a0, b1 = a1;   // current sub-tree weight, avg
      int32_t st_w[3], st_a0[3], st_a1[3], st_pw[3];
#pragma unroll
      for (int st = 0; st < 3; st++) {
        const int bit = 1 << st;
        const int32_t pw = lane_xor8(wa, bit);
        const int32_t p0 = lane_xor8(b0, bit), p1 = lane_xor8(b1, bit);
        st_w[st] = wa;
        st_a0[st] = b0;
        st_a1[st] = b1;
        st_pw[st] = pw;
        if (wa && pw) {
          b0 = (b0 + p0) >> 1;
          b1 = (b1 + p1) >> 1;
        } else if (pw) {
          b0 = p0;
          b1 = p1;
        }
        wa += pw;
      }
      // descend: the sub-tree containing this position is the RIGHT one
      // of a real pair -> its own ascent average, otherwise inherit
      int32_t d0 = on ? par2(ctx.dqp, par_par)[prow * 2] : 0;
      int32_t d1 = on ? par2(ctx.dqp, par_par)[prow * 2 + 1] : 0;
#pragma unroll
      for (int st = 2; st >= 0; st--) {
        const int bit = 1 << st;
        if ((t & bit) && st_w[st] && st_pw[st]) {
          d0 = st_a0[st];
          d1 = st_a1[st];
        }
      }
      if (has) {
        nq0 = d0 >> 4;
        nq1 = d1 >> 4;
        if (kRecon) {
          par2(ctx.dqp, cur_par)[crow * 2] = d0;
          par2(ctx.dqp, cur_par)[crow * 2 + 1] = d1;
        }
      }
    }

    // ---- butterfly weights + coefficients (mkWeightTree :742) ----------
    int32_t wl[3], wr[3];
    VC ca[3], cb[3];
    int32_t cw = w;
    if constexpr (REC) {
      // (of the weights only "both sides" / "right side only" is used below: flags)
#pragma unroll
      for (int st = 0; st < 3; st++) {
        const bool both = (rpk2 >> (21 + st)) & 1u, swap = (rpk2 >> (24 + st)) & 1u;
        wl[st] = swap ? 0 : 1;
        wr[st] = (both || swap) ? 1 : 0;
        ca[st] = A::coef(rf[(kSfCa + st) * rln]);
        cb[st] = A::coef(rf[(kSfCb + st) * rln]);
      }
      cw = (int32_t)((rpk2 >> (8 + t)) & 1u);  // this position holds a coefficient (`present`)
    } else {
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
    }

    // ---- inter-level prediction (tmc3/RAHT.cpp:1391-1432) --------------
    const bool inherit_dc = !e.is_root;
    const bool pred_in_level =
      on && inherit_dc && prm->raht_prediction_enabled_flag != 0;
    bool enable_pred = pred_in_level;
    int neigh_count = 0;
    VT pred[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = A::zero();

    bool do_search = false;
    if (pred_in_level) {
      if (ext && nchild == 1) {
        enable_pred = false;
        neigh_count = 19;
      } else if (par2(ctx.nneigh, par_par)[prow] < prm->raht_prediction_threshold0) {
        enable_pred = false;
      } else {
        do_search = true;
      }
    }
    // (group-uniform; other groups of the wave idle through the shuffles)
    int pn[3] = {-1, -1, -1};  // neighbour i = 1 + t + 8*slot
    if constexpr (REC) {
      // (the record pass searched wherever the level predicts: tmc3/RAHT.cpp:299-368)
      if (do_search) {
#pragma unroll
        for (int slot = 0; slot < 3; slot++)
          pn[slot] = rf[(kSfPn + slot) * rln];
      }
    } else if (GPCC_EXPERIMENTS && ctx.link_rec) {
      // round 5: the parent's record holds its 18 neighbours (raht_links.hpp) -- one load each instead of
      // a 12-step bisection; findNeighbour's window (tmc3/RAHT.cpp:272-293) is an index distance
      if (do_search) {
        const int64_t range = prm->raht_prediction_search_range;
        const int rj = ctx.link_lrec[j];
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          const int i = 1 + t + 8 * slot;
          if (i < 19 && (occ & neigh_mask(i)))
            pn[slot] = link_lookup(ctx.link_rec, rj, i, j, range);
        }
      }
    } else {
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
#ifdef GPCC_EXP_SEARCH2  // (experiment: what the search costs -- it is done twice, same result)
      {
        int l2[3] = {lo[0], lo[1], lo[2]}, h2[3] = {hi[0], hi[1], hi[2]};
        while (__any((l2[0] < h2[0]) | (l2[1] < h2[1]) | (l2[2] < h2[2]))) {
#pragma unroll
          for (int slot = 0; slot < 3; slot++) {
            const int mid = l2[slot] + ((h2[slot] - l2[slot]) >> 1);
            const int64_t kv = l2[slot] < h2[slot] ? pkey[mid] : 0;
            if (l2[slot] < h2[slot]) {
              if (kv < want[slot])
                l2[slot] = mid + 1;
              else
                h2[slot] = mid;
            }
          }
        }
        // the second search starts where the first one ended up saying it should (a dependency, no change)
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          int a2 = l2[slot];
          asm volatile("" : "+v"(a2));
          lo[slot] += a2 ^ l2[slot];
        }
      }
#endif
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
    {
      int found;
      if constexpr (REC) {
        found = (int)((rpk2 >> 16) & 31u);
      } else {
        found = (pn[0] >= 0) + (pn[1] >= 0) + (pn[2] >= 0);
        found = group8_sum(found);
      }
      if (do_search) {
        neigh_count = found + 1;
        if (neigh_count < prm->raht_prediction_threshold1)
          enable_pred = false;
      }
    }

    // ---- everything that does not wait: coefficient slots, quantisers,
    //      the encoder's source transform, parent-level prediction terms ---
    // ---- coefficient slot of this position (scanBlock :776-791) --------
    const uint32_t present = group8_bits(on && cw != 0) | (on ? 1u : 0u);
    // scan order 0,4,2,1,6,5,3,7 -> scan position of t
    const int spos = (0x74516230u >> (4 * t)) & 7;
    const uint32_t pscan = ((present >> 0) & 1) | (((present >> 4) & 1) << 1)
      | (((present >> 2) & 1) << 2) | (((present >> 1) & 1) << 3)
      | (((present >> 6) & 1) << 4) | (((present >> 5) & 1) << 5)
      | (((present >> 3) & 1) << 6) | (((present >> 7) & 1) << 7);
    const int rank = popc32(pscan & ((1u << spos) - 1));
    const bool coded = on && ((present >> t) & 1) && (t != 0 || !inherit_dc);
    // slice-relative coefficient index
    const int cidx = e.coeff_base
      + (inherit_dc ? (c0 - sc0) - pj + rank - 1 : rank);
    int32_t* __restrict__ cplane = ctx.coeffs + (size_t)pt0 * C + cidx;


    Quantizer qa[2] = {{1, 1}, {1, 1}};
    bool q_same = true;  // no AC offset here: the RDOQ quantiser is the coding quantiser
    if (coded) {
      int ac0 = 0, ac1 = 0;
      if (e.ac_layer < prm->num_ac_qp_layers && t) {
        ac0 = prm->ac_qp_offset[e.ac_layer][t - 1][0];
        ac1 = prm->ac_qp_offset[e.ac_layer][t - 1][1];
        q_same = (ac0 | ac1) == 0;
      }
      qpset_quantizers(prm, e.qp_layer, nq0 + ac0, nq1 + ac1, qa);
    }
    // the two normalisers of this position, settled before the wait (they
    // depend on the weight alone; on the chain they were a table look-up or
    // an irsqrt evaluation per hop): sqrt(w) for the prediction, the
    // (shift, 1/sqrt(w)) pair of scale_rsqrt for the reconstruction
    int32_t nrm_sq_i = 0, nrm_rs_i = 0, nrm_shift = 0;
    if constexpr (REC) {
      nrm_sq_i = rf[kSfNsq * rln];
      nrm_rs_i = rf[kSfNrs * rln];
      nrm_shift = (int)(rpk >> 27);
    } else if (!haar && w > 1) {
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
    if (kEnc && !REC) {
      // forward butterflies of the source (normalised first unless Haar:
      // scale_rsqrt, tmc3/RAHT.cpp:1474-1481)
      if (!haar && w > 1) {
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
            if (haar) {
              if constexpr (!A::kF64) {
                const int64_t hf = left ? oth - own : own - oth;
                src[k] = left ? own + ((hf >> (1 + kFpFrac)) << kFpFrac) : hf;
              }
            } else {
              src[k] = left ? A::mulc(oth, cb[st]) + A::mulc(own, ca[st])
                            : A::mulc(own, ca[st]) - A::mulc(oth, cb[st]);
            }
          } else if (swap) {
            src[k] = oth;
          }
        }
      }
    }
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
    Quantizer qr[2] = {{1, 1}, {1, 1}};
    if (kLossy && coded)
      qpset_quantizers(prm, e.qp_layer, nq0, nq1, qr);
    const typename A::Quant qra[2] = {A::quant(qr[0]), A::quant(qr[1])};
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
    //      not depend on this launch is settled here ------------------------
    // Which term a (lane, neighbour) pair contributes is decided by the tree
    // alone: the neighbour parent's value (absent or later block, or the
    // needed child position unoccupied), a child written by the prepass
    // (single-child parent), or a child reconstructed by THIS launch -- only
    // the last kind is waited for, through its 16-byte mailbox granule.
    // lane t owns neighbours i = 1 + t + 8*slot (the lanes that searched them)
    int nb_c0[3] = {0, 0, 0};
    uint32_t nb_occ[3] = {0, 0, 0};
    int nb_single[3] = {0, 0, 0};
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
        if constexpr (REC) {
          nb_c0[slot] = rf[(kSfNbc0 + slot) * rln];
          nb_occ[slot] = (rpk >> (8 * slot)) & 0xffu;  // (zero for a neighbour behind this block)
          nb_single[slot] = (int)((rpk >> (24 + slot)) & 1u);
        } else if (q < j) {  // processed before this block: its children count
          const int qc0 = tv.fc[li + 1][q];
          nb_c0[slot] = qc0;
          nb_occ[slot] = ctx.pocc[q];
          nb_single[slot] = ext && tv.fc[li + 1][q + 1] - qc0 == 1;
        }
      }
    }
    uint32_t pend = 0;       // neighbours whose child granule is awaited
    int32_t nrow12[12];      // row of that child in rec / mbox
    // A neighbour block claimed by THIS wavefront in this round (about 70 % of
    // the hops on the longest chains: Morton-adjacent blocks) hands its
    // children over through registers instead of a memory round trip:
    // inw = awaited neighbours of that kind, wsrc = their group, 3 bits each.
    uint32_t inw = 0, wsrc_a = 0, wsrc_b = 0;
    int jg[8];
#pragma unroll
    for (int g = 0; g < 8; g++)
      jg[g] = __shfl(on ? j : -1, g << 3);
#pragma unroll
    for (int i12 = 0; i12 < 12; i12++) {
      nrow12[i12] = 0;
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
            // copied by the prepass launch: an ordinary load
            const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
            for (int k = 0; k < C; k++) {
              const VT cv = A::from_i64(par2(ctx.rec, cur_par)[nrow * C + k]);
              in_range = in_range && A::below(cv, A::kRecLimit);
              pred[k] += A::muli(cv, mul);
            }
          } else {
            nrow12[i12] = (int32_t)nrow;
            pend |= 1u << i12;
            int pg = -1;
#pragma unroll
            for (int g = 0; g < 7; g++)
              pg = q == jg[g] ? g : pg;
            if (pg >= 0) {
              inw |= 1u << i12;
              if (i12 < 10)
                wsrc_a |= (uint32_t)pg << (3 * i12);
              else
                wsrc_b |= (uint32_t)pg << (3 * (i12 - 10));
            }
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
    const auto mrsrc = __builtin_amdgcn_make_buffer_rsrc(
      ctx.mbox, 0, (int)((size_t)tv.n_total * C * 16), 0x00020000);

    // ---- the reference frame's block (tmc3/RAHT.cpp:1322-1347, 1533-1545): where it exists it is the
    //      prediction of every coefficient of the block, and nothing of this level is waited for ---------
    bool use_inter = false;
    VT ipin[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      ipin[k] = A::zero();
    if constexpr (INTER) {
      if (ctx.inter.blocks) {
        bool node;
        int64_t pin[C];
        if (ctx.inter.hkey)
          inter_block_haar<C>(ctx.inter, on ? tv.key[li + 1][j] : 0, t, on, &node, pin);
        else
          inter_block<C>(ctx.inter, on ? tv.key[li + 1][j] : 0, t, on, lut, &node, pin);
        use_inter = node;
#pragma unroll
        for (int k = 0; k < C; k++) {
          ipin[k] = A::from_i64(pin[k]);
          in_range = in_range && (!node || A::below(ipin[k], A::kFwdLimit));
        }
        if (use_inter) {
          pend = 0;
          inw = 0;
          enable_pred = on;
        }
      }
    }

    // ---- the staged dependency loop -------------------------------------
    // stage 0: waiting for neighbour blocks   -> (P) predict + transform,
    //          results cached in registers
    // stage 1: waiting for the RDOQ state L (lossy encoder only)
    // stage 3: committed
    // A waiting iteration costs the polls and a handful of tests; (P) runs when a group
    // becomes neighbour-ready, the RDOQ evaluation (Z) when a group has its descriptors or
    // learns the incoming state, the commit part (W) when one can commit.  Children of
    // blocks of THIS wavefront are handed over through the wavefront's LDS mailbox.
    int stage = on ? 0 : 3;
    unsigned spins = 0;
    VT pt[C];               // transformed prediction of this position
    int32_t qc[C];          // tentative quantised coefficients (encoder) / coded ones (decoder)
    uint32_t dr = kDescZero;  // RDOQ descriptor of rank t (lossy encoder)
    // RDOQ of the block (group-uniform): zs 0 = descriptors not there yet, 1 = the decisions
    // wait for the incoming last reset, 2 = decided (zero_r valid)
    int zs = 0;
    bool zero_r = false;
    uint32_t dmask = 0;     // definite resets of the block, by rank
    bool lin_exact = false; // lin is the incoming last reset itself (not only a bound that settles the block)
    int need = 0;           // reach of the block's thresholds before its first coefficient
    int look = wi - 1;      // look-back cursor
    int outk = 0, outv = -1;   // outgoing RDOQ state: 0 unknown, 1 transparent, 2 final (= outv)
    unsigned long long done = 0;  // children of this round whose value lies in the mailbox (wave-uniform)
    int poll_last = 31;           // the granule polled last (31: none yet -- start with the lowest awaited one)
    unsigned long long done_seen = ~0ull;  // `done` when the mailbox loop last ran
    int cur_slot = -1, cur_pwc = 0;        // the granule this lane polls: slot, weight, row -- looked up when the slot changes
    int32_t cur_row = 0;
    int cur_mul = 0;
#pragma unroll
    for (int k = 0; k < C; k++) {
      pt[k] = A::zero();
      // decoder: the coded coefficients are input -- fetched here, not on the chain
      qc[k] = (!kEnc && coded) ? cplane[(size_t)k * n_s] : 0;
    }
    prof.loop_begin();
    while (__any(stage != 3)) {
      bool progressed = false;
      prof.iter_begin();
      // ---- (X) awaited children of blocks of this wavefront: the mailbox ----
      // (only when the mailbox has gained a child since the loop last ran: nothing else lets it consume one, and a waiting
      // iteration -- a round waits for ~100 of them -- is then the poll below and a handful of tests)
      if (!GPCC_SUB_IDLE_FAST || done != done_seen) {
        done_seen = done;
        uint32_t mi = stage == 0 ? (pend & inw) : 0u;
        while (__any(mi != 0)) {
          const bool act = mi != 0;
          const int slot = act ? __ffs(mi) - 1 : 0;
          mi &= mi - 1;
          const uint32_t pg = (slot < 10 ? wsrc_a >> (3 * slot) : wsrc_b >> (3 * (slot - 10))) & 7u;
          // occuShift of findNeighbours (tmc3/RAHT.cpp:377), three bits each
          constexpr unsigned long long kShifts = 06ull | 05ull << 3 | 04ull << 6 | 03ull << 9 | 02ull << 12
            | 01ull << 15 | 03ull << 18 | 01ull << 21 | 02ull << 24 | 01ull << 27 | 02ull << 30 | 03ull << 33;
          const int sh = (int)((kShifts >> (3 * slot)) & 7);
          const int srcl = (int)(pg << 3) | ((slot < 9 ? t + sh : t - sh) & 7);
          if (act && ((done >> srcl) & 1)) {
            int pwc = 0;
#pragma unroll
            for (int i12 = 0; i12 < 12; i12++)
              pwc = slot == i12 ? pwc12[i12] : pwc;
            const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
            for (int k = 0; k < C; k++)
              pred[k] += A::muli(__builtin_bit_cast(VT, wm[srcl * C + k]), mul);
            pend &= ~(1u << slot);
            inw &= ~(1u << slot);
          }
        }
      }
      // ---- (X) the others: the granule is data and flag at once -----------
      // One granule per lane and iteration -- the lowest awaited one -- so an
      // iteration costs ONE memory round trip however many different
      // neighbours the 64 lanes wait for (polling slot after slot made it
      // as many round trips as there were slots in use).
      // (Skipping the poll in an iteration that has a group ready to compute -- so that a chain
      // inside the wavefront never stalls on a memory round trip -- was measured: 7.22 / 4.17 ms
      // against 6.98 / 3.95, arrivals from other wavefronts are what the frame waits for.)
      const uint32_t pm = stage == 0 ? (pend & ~inw) : 0u;
#if GPCC_SUB_POLL_N > 1
      // (experiment) the GPCC_SUB_POLL_N lowest awaited granules of a lane side by side: neighbours produced by ONE earlier
      // round arrive together, and one granule per iteration consumes them a memory round trip apart
      if (pm) {
        uint32_t pmm = pm;
        int slot_[GPCC_SUB_POLL_N], pwc_[GPCC_SUB_POLL_N];
        bool act_[GPCC_SUB_POLL_N];
        u32x4 g[GPCC_SUB_POLL_N][C];
#pragma unroll
        for (int u = 0; u < GPCC_SUB_POLL_N; u++) {
          act_[u] = pmm != 0;
          slot_[u] = act_[u] ? __ffs(pmm) - 1 : __ffs(pm) - 1;
          pmm &= pmm - 1;
          int32_t row = 0;
          pwc_[u] = 0;
#pragma unroll
          for (int i12 = 0; i12 < 12; i12++) {
            if (slot_[u] == i12) {
              row = nrow12[i12];
              pwc_[u] = pwc12[i12];
            }
          }
#pragma unroll
          for (int k = 0; k < C; k++)
            g[u][k] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, (row * C + k) * 16, 0, /*sc1*/ 16);
        }
#pragma unroll
        for (int u = 0; u < GPCC_SUB_POLL_N; u++) {
          bool ok = act_[u];
#pragma unroll
          for (int k = 0; k < C; k++)
            ok = ok && g[u][k].z == ctx.mtag;
          if (ok) {
            const int mul = ext ? pwc_[u] : (pwc_[u] << kFpFrac);
#pragma unroll
            for (int k = 0; k < C; k++)
              pred[k] += A::muli(__builtin_bit_cast(VT, ((uint64_t)g[u][k].y << 32) | g[u][k].x), mul);
            pend &= ~(1u << slot_[u]);
          }
        }
      }
#else
      if (pm) {
#if GPCC_SUB_POLL_RR
        // round robin over the awaited granules: the one behind the last polled.  Polling the LOWEST awaited one until it
        // arrives makes every other neighbour -- long there -- wait its turn behind a late one, an iteration each after it
        // has arrived; in turn they are consumed while the late one is still out (the sums are exact: order is free)
        const uint32_t above = pm & ~((2u << poll_last) - 1u);
        const int slot = __ffs(above ? above : pm) - 1;
        poll_last = slot;
#else
        const int slot = __ffs(pm) - 1;
#endif
        int32_t row = 0;
        int pwc = 0;
        if (GPCC_SUB_IDLE_FAST && !GPCC_SUB_POLL_RR) {
          if (slot != cur_slot) {
            cur_slot = slot;
#pragma unroll
            for (int i12 = 0; i12 < 12; i12++) {
              if (slot == i12) {
                cur_row = nrow12[i12];
                cur_pwc = pwc12[i12];
              }
            }
          }
          row = cur_row;
          pwc = cur_pwc;
        } else {
#pragma unroll
          for (int i12 = 0; i12 < 12; i12++) {
            if (slot == i12) {
              row = nrow12[i12];
              pwc = prm->pred_weight_child[i12];
            }
          }
        }
#if GPCC_SUB_IDLE_SPIN
        cur_mul = ext ? pwc : (pwc << kFpFrac);
      }
      {
        // Nothing but an arrival from another wavefront lets this one go on -- no group waits for the zero-run state,
        // none is ready to predict, the mailbox has been looked at: poll in place.  (The waiting iteration is what a
        // round spends ~100 of, three wavefronts per SIMD at a time: every instruction of it is also taken from the
        // issue slots of the wavefront that has work -- profiles/r06_idle_ab.txt.)
        const bool spin_ok = !__any(stage == 1) && !__any(stage == 0 && !group8_any(stage == 0 && pend));
        u32x4 g[C];
        bool ok = false;
        bool expired = false;
        for (;;) {
          if (pm) {
#pragma unroll
            for (int k = 0; k < C; k++)
              g[k] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, (cur_row * C + k) * 16, 0, /*sc1*/ 16);
            ok = true;
#pragma unroll
            for (int k = 0; k < C; k++)
              ok = ok && g[k].z == ctx.mtag;
          }
          if (!spin_ok || __any(ok) || !__any(pm != 0))
            break;
          prof.idle();
          if (++spins > (1u << 21)) {
            expired = true;
            break;
          }
          __builtin_amdgcn_s_sleep(GPCC_SUB_SLEEP);
        }
        if (expired) {
          if (lane == 0)
            atomicExch(ctx.error, 1);  // fail loudly instead of hanging the GPU
          break;
        }
        if (ok) {
          // (a granule carries the value in the launch's arithmetic: every reader is this launch)
#pragma unroll
          for (int k = 0; k < C; k++)
            pred[k] += A::muli(__builtin_bit_cast(VT, ((uint64_t)g[k].y << 32) | g[k].x), cur_mul);
          pend &= ~(1u << cur_slot);
        }
      }
#else
        u32x4 g[C];
#pragma unroll
        for (int k = 0; k < C; k++)
          g[k] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, (row * C + k) * 16, 0, /*sc1*/ 16);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < C; k++)
          ok = ok && g[k].z == ctx.mtag;
        if (ok) {
          // (a granule carries the value in the launch's arithmetic: every reader is this launch)
          const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
          for (int k = 0; k < C; k++)
            pred[k] += A::muli(__builtin_bit_cast(VT, ((uint64_t)g[k].y << 32) | g[k].x), mul);
          pend &= ~(1u << slot);
        }
      }
#endif
#endif
      prof.mark<0>();
      const bool blocked = group8_any(stage == 0 && pend);
      const bool nready = stage == 0 && !blocked;

      if (__any(nready)) {
        progressed = true;
        // ---- (P) normalise the prediction, transform -----------------------
        prof.template enter<0>();
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
        if constexpr (INTER) {
          if (use_inter) {
#pragma unroll
            for (int k = 0; k < C; k++)
              pw_[k] = ipin[k];
          }
        }
        prof.template sub<0>();
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
              // an all-zero coefficient never resets; whether RDOQ "zeroes" it only matters if the AC-offset
              // quantiser made the tentative value non-zero -- otherwise it must not make the block wait for L,
              // and its threshold (a double division and two exact corrections: ~700 cycles on the hop) is never
              // looked at: on a smooth field that is nearly every coefficient, and the branch is skipped wave-wide
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
        prof.template sub<1>();
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
      // (tmc3/RAHT.cpp:1618-1669 restated in raht_rdoq.hpp: rank r resets iff it is a definite
      // reset, or run-dependent with a reset in [r - thr, r - 1]; the block's resets for an incoming
      // last reset l0 are the least fixed point, reached in as many sweeps as failures chain.)
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
        // sweeps run in wave-uniform control flow; lanes of groups that do not take part pass act = false
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
              lin = ctx.slice_l[((li + 1) & 1) * tv.num_slices + s];
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
                  __hip_atomic_store(
                    &ctx.rdoq_state[wi], ((unsigned long long)epoch << 48) | (2ull << 32) | (uint32_t)outv,
                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
          }
        }
        if (__any(stage == 1 && zs == 1)) {
          // ---- the incoming L: predecessors in memory (a wavefront's first groups), 8 per step;
          // only a reset within `need` coefficients before the block matters ----
          const bool wantm = stage == 1 && zs == 1 && !lin_known;
          if (__any(wantm)) {
            const int kk = look - t;
            const int kc = kk < 0 ? 0 : kk;
            unsigned long long sv = 0;
            int wl_k = 0;
            if (wantm) {
              sv = __hip_atomic_load(&ctx.rdoq_state[kc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              wl_k = ctx.worklist[kc];
            }
            const bool boundary = kk < 0 || wl_k < sp0;
            const bool cur_ep = (sv >> 48) == (unsigned long long)epoch;
            // kind: 0 pending, 1 reset-free (value = its first coefficient),
            // 2 final (value = L), 3 slice start, 5 far enough
            int kind = boundary ? 3 : (cur_ep ? (int)((sv >> 32) & 0xffff) : 0);
            const int val = (int)(uint32_t)sv;
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
                lin = ctx.slice_l[((li + 1) & 1) * tv.num_slices + s];  // as the previous level left it
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
            // (nothing but reset-free blocks before this one in the round: the state the claim's previous
            // round left, in registers)
#if GPCC_EXPERIMENTS
            const bool from_carry = w2 && nt == 0 && carry_known;
            const bool found = (w2 && nt != 0 && pk == 2) || from_carry;
            if (found) {
              lin = from_carry ? carry_l : pv;
              lin_known = lin_exact = true;
            }
#else
            const bool found = w2 && nt != 0 && pk == 2;
            if (found) {
              lin = pv;
              lin_known = lin_exact = true;
            }
#endif
            if (!__any(found))
              break;
          }
        }
        can = stage == 1 && zs == 2;
        if (can && t == 0) {
          // what this block does to L: its last reset (or the known state passing through), or
          // nothing (then the word carries the block's first coefficient index for the walk)
          const unsigned long long ep = (unsigned long long)epoch << 48;
          const unsigned long long word =
            outk >= 2 ? ep | (2ull << 32) | (uint32_t)outv : ep | (1ull << 32) | (uint32_t)cfirst;
          __hip_atomic_store(&ctx.rdoq_state[wi], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (outk >= 2)
            atomicMax(&ctx.slice_l[(li & 1) * tv.num_slices + s], outv);  // L carried to the next level
        }
      }

      prof.mark<2>();
      if (__any(can)) {
        progressed = true;
        // ---- (W) coefficients, DC, inverse transform, commit ---------------
        prof.template enter<1>();
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
        prof.template sub<2>();
        // children of the committing groups: the value later launches read
        // (plain store) and the mailbox granule later blocks of THIS launch
        // poll (one 16-byte write-through store: value + tag, untorn)
        if (can && has) {
          // the granules first: they are what the next hop of the chain waits for
          VT vn[C];
#pragma unroll
          for (int k = 0; k < C; k++) {
            VT v = pw_[k];
            if (!haar && w > 1)
              v = A::mulc(A::shr(v, nrm_shift), nrm_rs);
            v = ext ? v : A::round_int(v);
            vn[k] = v;
            const uint64_t vb = __builtin_bit_cast(uint64_t, v);
            wm[lane * C + k] = vb;  // read by later groups of this wavefront once `done` says so
            const u32x4 gr = {(uint32_t)vb, (uint32_t)(vb >> 32), ctx.mtag, 0u};
#if GPCC_SUB_DSTORE
            // (first into this XCD's L2, where the pollers of the same XCD find it at once; then through to memory)
            __builtin_amdgcn_raw_buffer_store_b128(gr, mrsrc, (int)((crow * C + k) * 16), 0, 0);
#endif
            __builtin_amdgcn_raw_buffer_store_b128(gr, mrsrc, (int)((crow * C + k) * 16), 0, /*sc1*/ 16);
          }
          prof.template sub<3>();
#pragma unroll
          for (int k = 0; k < C; k++) {
            par2(ctx.rec_us, cur_par)[crow * C + k] = A::to_i64(ext ? pw_[k] : A::round_int(A::muli(pw_[k], 4)));
            par2(ctx.rec, cur_par)[crow * C + k] = A::to_i64(vn[k]);
          }
          par2(ctx.nneigh, cur_par)[crow] = inherit_dc ? neigh_count : 19;
          prof.template sub<4>();
        }
        if (can)
          stage = 3;
      }

      done |= __ballot(can && has);
      prof.mark<3>();
      if (!progressed) {
        prof.idle();
        if (++spins > (1u << 21)) {
          if (lane == 0)
            atomicExch(ctx.error, 1);  // fail loudly instead of hanging the GPU
          break;
        }
        __builtin_amdgcn_s_sleep(GPCC_SUB_SLEEP);
      }
    }
    prof.round_end(lane, li);
#if GPCC_EXPERIMENTS
    if (kLossy && claim_rounds > 1) {
      // what this round leaves for the claim's next one: the state behind its last block that knows it
      // (every later block of the round is reset-free); a round of reset-free blocks that never learnt the
      // incoming state leaves none -- the next round then walks the words in memory as any other claim does
      const unsigned long long fin = __ballot(outk == 2) & 0x0101010101010101ull;
      if (fin) {
        carry_l = __shfl(outv, 63 - __clzll((long long)fin));
        carry_known = true;
      } else if (__any(outk == 1)) {
        carry_known = false;
      }
    }
   }
#endif
  }
  // ArithF64: a value left the range in which doubles are exact -- the sticky word stops every
  // later kernel of the call (the source attributes stay intact) and the call is redone with
  // ArithI64 (host tier) or reports GPCC_ERR_RANGE (device tier)
  if (A::kF64 && __any(!in_range) && lane == 0)
    atomicCAS(ctx.error, 0, 3);
}

}  // namespace gpcc
