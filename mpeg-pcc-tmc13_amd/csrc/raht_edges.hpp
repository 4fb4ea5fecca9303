// raht_edges.hpp -- the non-block parts of the RAHT pipeline:
//   ascend_*  : integer-Haar low-pass values and region-QP averages, which
//               are NOT associative and therefore computed level by level
//               exactly as reduceUnique/reduceLevel do
//               (tmc3/RAHT.cpp:108-205);
//   finish    : duplicate-point tail (tmc3/RAHT.cpp:1840-1964), the
//               single-point shortcut (:998-1017) and the rounded
//               write-back of the reconstruction (:1967-1975).
#pragma once

#include "raht_common.hpp"
#include "raht_levels.hpp"

namespace gpcc {

struct AscendCtx {
  TreeView tv;
  const int32_t* attrs;    // [N][C] source attributes (Haar encoder) or null
  const int32_t* qp_off;   // [N][2] or null
  int32_t* const* haar_lf; // [nlev] -> [M][C]
  int32_t* const* asc_qp;  // [nlev] -> [M][2]
  int32_t* dup_hf;         // [N][C] Haar difference of each duplicate point
  int32_t* dqp_root;       // dqp[1]: the root block's parent qp lives here
  int32_t li;              // level being produced
};

// level 0 from the points (reduceUnique)
template<int C>
__global__ __launch_bounds__(256) void
ascend_leaf_kernel(AscendCtx cx)
{
  const TreeView& tv = cx.tv;
  if (tree_failed(tv))
    return;
  const int m = tv.soff[0][tv.num_slices];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m;
       j += gridDim.x * blockDim.x) {
    const int f0 = tv.fp[0][j], f1 = tv.fp[0][j + 1];
    if (cx.asc_qp) {
      cx.asc_qp[0][(size_t)j * 2] = cx.qp_off[(size_t)f0 * 2] * 16;
      cx.asc_qp[0][(size_t)j * 2 + 1] = cx.qp_off[(size_t)f0 * 2 + 1] * 16;
    }
    if (cx.haar_lf) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        int32_t acc = cx.attrs[(size_t)f0 * C + k];
        for (int i = f0 + 1; i < f1; i++) {
          const int32_t d = (int32_t)((uint32_t)cx.attrs[(size_t)i * C + k] - (uint32_t)acc);
          cx.dup_hf[(size_t)i * C + k] = d;
          acc = (int32_t)((uint32_t)acc + (uint32_t)(d >> 1));
        }
        cx.haar_lf[0][(size_t)j * C + k] = acc;
      }
    }
  }
}

// level li from level li-1: three binary merges z, y, x (reduceLevel)
template<int C>
__global__ __launch_bounds__(256) void
ascend_level_kernel(AscendCtx cx)
{
  GPCC_VGPR_FLOOR_64();
  const TreeView& tv = cx.tv;
  if (tree_failed(tv))
    return;
  const int li = cx.li;
  const int m = tv.soff[li][tv.num_slices];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m;
       j += gridDim.x * blockDim.x) {
    const int c0 = tv.fc[li][j], c1 = tv.fc[li][j + 1];
    int32_t w[8], a[8][C], q[8][2];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      w[i] = 0;
      q[i][0] = q[i][1] = 0;
#pragma unroll
      for (int k = 0; k < C; k++)
        a[i][k] = 0;
    }
    for (int ch = c0; ch < c1; ch++) {
      const int idx = (int)(tv.key[li - 1][ch] & 7);
      // select by idx with a compare chain to keep the arrays in registers
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (i == idx) {
          w[i] = tv.fp[li - 1][ch + 1] - tv.fp[li - 1][ch];
          if (cx.asc_qp) {
            q[i][0] = cx.asc_qp[li - 1][(size_t)ch * 2];
            q[i][1] = cx.asc_qp[li - 1][(size_t)ch * 2 + 1];
          }
          if (cx.haar_lf) {
#pragma unroll
            for (int k = 0; k < C; k++)
              a[i][k] = cx.haar_lf[li - 1][(size_t)ch * C + k];
          }
        }
      }
    }
#pragma unroll
    for (int step = 1; step < 8; step <<= 1) {
#pragma unroll
      for (int l = 0; l < 8; l += 2 * step) {
        const int r = l + step;
        if (!w[r])
          continue;
        if (!w[l]) {
          w[l] = w[r];
          q[l][0] = q[r][0];
          q[l][1] = q[r][1];
#pragma unroll
          for (int k = 0; k < C; k++)
            a[l][k] = a[r][k];
        } else {
          w[l] += w[r];
          q[l][0] = (q[l][0] + q[r][0]) >> 1;
          q[l][1] = (q[l][1] + q[r][1]) >> 1;
#pragma unroll
          for (int k = 0; k < C; k++) {
            const int32_t d = (int32_t)((uint32_t)a[r][k] - (uint32_t)a[l][k]);
            a[l][k] = (int32_t)((uint32_t)a[l][k] + (uint32_t)(d >> 1));
          }
        }
        w[r] = 0;
      }
    }
    if (cx.asc_qp) {
      cx.asc_qp[li][(size_t)j * 2] = q[0][0];
      cx.asc_qp[li][(size_t)j * 2 + 1] = q[0][1];
    }
    if (cx.haar_lf) {
#pragma unroll
      for (int k = 0; k < C; k++)
        cx.haar_lf[li][(size_t)j * C + k] = a[0][k];
    }
  }
}

// the root keeps its ascent average as descent qp
__global__ void
qp_root_kernel(AscendCtx cx, const SliceSched* sched)
{
  const TreeView& tv = cx.tv;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= tv.num_slices || tree_failed(tv))
    return;
  const int top = sched[s].top_level;
  const int node = tv.soff[top][s];
  const int64_t row = tv.pt_off[s];
  cx.dqp_root[row * 2] = cx.asc_qp[top][(size_t)node * 2];
  cx.dqp_root[row * 2 + 1] = cx.asc_qp[top][(size_t)node * 2 + 1];
}

struct FinishCtx {
  TreeView tv;
  const gpcc_raht_params* params;
  const SliceSched* sched;
  const int32_t* attr_prefix;
  const int32_t* const* haar_lf;
  const int32_t* dup_hf;
  const int32_t* const* asc_qp;
  const int32_t* qp_off;
  const int64_t* rec[2];
  const int32_t* dqp[2];
  // compact level pass (cx_level.hpp): the leaves' values live in value slots
  const int64_t* slot_rec;  // [2N][C] or null
  int32_t slot_f64;         // the slots hold doubles' bit patterns (the pass ran in ArithF64)
  const uint32_t* hold0;    // [M0] slot of every leaf
  int32_t* attrs;   // in: source (encoder), out: reconstruction
  int32_t* coeffs;
  int32_t encoder;
  const SharedLut* lut;
};

#ifndef GPCC_FIN_VAR
#define GPCC_FIN_VAR 0
#endif
#if GPCC_FIN_VAR == 2
// [0] checks, [1] mismatching words after the barrier, [2] at the end, then records of
// {phase << 16 | word, workgroup, LDS value, global value}
__device__ unsigned int g_fin_dbg[4 + 4 * 60];
__device__ __forceinline__ void
fin_check_lut(const SharedLut* s, const SharedLut* g, int phase)
{
  const volatile uint32_t* l = reinterpret_cast<const volatile uint32_t*>(s);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
  if (threadIdx.x == 0)
    atomicAdd(&g_fin_dbg[0], 1u);
  for (int i = threadIdx.x; i < (int)(sizeof(SharedLut) / 4); i += blockDim.x) {
    const uint32_t a = l[i], b = __builtin_nontemporal_load(&src[i]);
    if (a != b) {
      const unsigned k = atomicAdd(&g_fin_dbg[1 + phase], 1u);
      if (k < 60) {
        g_fin_dbg[4 + 4 * k + 0] = (unsigned)phase << 16 | (unsigned)i;
        g_fin_dbg[4 + 4 * k + 1] = blockIdx.x;
        g_fin_dbg[4 + 4 * k + 2] = a;
        g_fin_dbg[4 + 4 * k + 3] = b;
      }
    }
  }
}
#endif

template<int C>
#if GPCC_FIN_VAR == 5
[[clang::optnone]] __attribute__((noinline))
#endif
__global__ __launch_bounds__(256) void
finish_kernel(FinishCtx cx)
{
  GPCC_VGPR_FLOOR_64();
  if (tree_failed(cx.tv))
    return;
  // The tables are read where lut_init_kernel left them (3 KB, cache resident; the chains of
  // duplicates that use them are rare).  The form with an LDS copy staged per workgroup, as in the
  // level kernels, gave WRONG coefficients of long duplicate chains on the MI355X in large batches
  // (round 3).  Round 4 (profiles/r04_finish_lds_root_cause.txt): the staged copy is bit-identical to
  // the global tables, and THE SAME MACHINE CODE is right or wrong depending only on the wavefront's
  // register allocation in the kernel descriptor -- 56 VGPRs, all used, as the compiler arrives at for
  // that form: 70-80 % of the runs of the pinned batches wrong, every lane of a wavefront in the same
  // loop iterations; 64 / 72 / 80 registers, no instruction changed: none (0 / 198).  Nothing at
  // source level is wrong; an LDS form, if ever wanted, pads its allocation (GPCC_FIN_VAR 8 below).
#if GPCC_FIN_VAR >= 1
  // experiment builds (tools/fin_lds_experiment.py): the LDS form that failed in round 3 and
  // variants of it; GPCC_FIN_VAR 2 adds checks of the staged copy against the global tables
  // (right behind the barrier and when the workgroup's threads leave), counted in g_fin_dbg
  __shared__ SharedLut lut_s;
  load_lut(&lut_s, cx.lut);
  const SharedLut& lut = lut_s;
#if GPCC_FIN_VAR == 2
  fin_check_lut(&lut_s, cx.lut, 0);
#endif
#if GPCC_FIN_VAR == 8
  // the same machine code in a 64-register allocation instead of the 56 the compiler arrives at
  asm volatile("" ::: "v63");
#endif
#else
  const SharedLut& lut = *cx.lut;
#endif
  const TreeView& tv = cx.tv;
  const gpcc_raht_params* __restrict__ prm = cx.params;
  const bool haar = prm->integer_haar_enable_flag != 0;
  const bool ext = prm->raht_extension != 0;
  const int m = tv.soff[0][tv.num_slices];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m;
       j += gridDim.x * blockDim.x) {
    const int s = find_slice(tv.soff[0], tv.num_slices, j);
    const SliceSched* sc = &cx.sched[s];
    const int pt0 = tv.pt_off[s];
    const int n_s = tv.pt_off[s + 1] - pt0;
    const int f0 = tv.fp[0][j], f1 = tv.fp[0][j + 1];
    const int weight = f1 - f0;
    const int jl = j - tv.soff[0][s];
    int32_t* __restrict__ co = cx.coeffs + (size_t)pt0 * C;

    if (n_s == 1) {
      // single point (tmc3/RAHT.cpp:998-1017): quantise directly, layer 0
      Quantizer q[2];
      qpset_quantizers(
        prm, 0, cx.qp_off ? cx.qp_off[(size_t)f0 * 2] : 0,
        cx.qp_off ? cx.qp_off[(size_t)f0 * 2 + 1] : 0, q);
#pragma unroll
      for (int k = 0; k < C; k++) {
        int64_t coeff;
        if (cx.encoder) {
          coeff = quantize(q[k ? 1 : 0], (int64_t)cx.attrs[(size_t)f0 * C + k] << 8);
          co[k] = (int32_t)coeff;
        } else {
          coeff = co[k];
        }
        cx.attrs[(size_t)f0 * C + k] = (int32_t)dequantize(q[k ? 1 : 0], coeff);
      }
      continue;
    }

    const bool any_level = sc->num_unique > 1;
    const int par = sc->final_parity;
    const int64_t row = (int64_t)pt0 + jl;
    int64_t rec[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      rec[k] = !any_level ? 0
        : (cx.slot_rec ? cx.slot_rec[(size_t)(cx.hold0[j] & kCxSlotMask) * C + k]
                       : cx.rec[par][row * C + k]);
    if (cx.slot_rec && cx.slot_f64) {
#pragma unroll
      for (int k = 0; k < C; k++)
        rec[k] = (int64_t)__builtin_bit_cast(double, rec[k]);
    }

    if (weight == 1) {
#pragma unroll
      for (int k = 0; k < C; k++)
        cx.attrs[(size_t)f0 * C + k] =
          ext ? (int32_t)((rec[k] + kFpHalf) >> kFpFrac) : (int32_t)rec[k];
      continue;
    }

    // ---- duplicates: a chain of (w, 1) two-point transforms ------------
    int nq0 = 0, nq1 = 0;
    if (cx.asc_qp) {
      if (any_level) {
        nq0 = cx.dqp[par][row * 2] >> 4;
        nq1 = cx.dqp[par][row * 2 + 1] >> 4;
      } else {
        nq0 = cx.asc_qp[0][(size_t)j * 2] >> 4;
        nq1 = cx.asc_qp[0][(size_t)j * 2 + 1] >> 4;
      }
    }
    Quantizer q[2];
    qpset_quantizers(prm, sc->final_qp_layer, nq0, nq1, q);
    const int64_t sq = sqrt_weight(weight, lut);
    int64_t attr_sum[C], rec_dc[C];
#pragma unroll
    for (int k = 0; k < C; k++) {
      attr_sum[k] = 0;
      if (cx.encoder) {
        const int32_t v = haar
          ? cx.haar_lf[0][(size_t)j * C + k]
          : (int32_t)((uint32_t)cx.attr_prefix[(size_t)f1 * C + k]
                      - (uint32_t)cx.attr_prefix[(size_t)f0 * C + k]);
        attr_sum[k] = fp_from_int(v);
      }
      rec_dc[k] = ext ? rec[k] : fp_from_int(rec[k]);
      if (!haar)
        rec_dc[k] = fp_mul(rec_dc[k], sq);
    }
    // coefficients of this node follow all level coefficients and the
    // tails of the earlier duplicated leaves
    // (a slice with one unique position codes no level coefficient at all)
    int cidx = (any_level ? sc->num_unique : 0) + (f0 - pt0) - jl;
    for (int wv = weight - 1; wv > 0; wv--, cidx++) {
      int64_t a = 0, b = 0;
      if (!haar) {
#if GPCC_FIN_VAR == 4
        int one = 1;  // opaque: nothing of the (w, 1) butterfly is hoisted out of the loops
        asm volatile("" : "+v"(one));
        raht_coeffs(wv, one, lut, &a, &b);
#else
        raht_coeffs(wv, 1, lut, &a, &b);
#endif
      }
#pragma unroll
      for (int k = 0; k < C; k++) {
        const Quantizer qk = q[k ? 1 : 0];
        int64_t t0, t1, coeff;
        if (cx.encoder) {
          if (haar) {
            t1 = fp_from_int(cx.dup_hf[(size_t)(f0 + wv) * C + k]);
            attr_sum[k] -= t1 >> 1;
            t1 += attr_sum[k];
            t0 = attr_sum[k];
            t1 = t1 - t0;  // HaarKernel::fwdTransform high-pass
          } else {
#if GPCC_FIN_VAR == 6   // the source attribute read past the CU's L1
            t1 = fp_from_int(__hip_atomic_load(&cx.attrs[(size_t)(f0 + wv) * C + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#else
            t1 = fp_from_int(cx.attrs[(size_t)(f0 + wv) * C + k]);
#endif
            attr_sum[k] -= t1;
            t0 = scale_rsqrt(attr_sum[k], wv, lut);
            t1 = fp_mul(t1, a) - fp_mul(b, t0);
          }
          coeff = quantize(qk, fp_round(t1) * 256);
          co[(size_t)k * n_s + cidx] = (int32_t)coeff;
        } else {
          coeff = co[(size_t)k * n_s + cidx];
        }
        t1 = fp_from_int(dequantize(qk, coeff));
        t0 = rec_dc[k];
        if (haar) {
          const int64_t left = t0 - ((t1 >> (1 + kFpFrac)) << kFpFrac);
          t1 = t1 + left;
          t0 = left;
        } else {
          const int64_t lf = t0, hf = t1;
          t0 = fp_mul(lf, a) - fp_mul(b, hf);
          t1 = fp_mul(lf, b) + fp_mul(a, hf);
        }
        rec_dc[k] = t0;
        const int64_t o1 = ext ? t1 : fp_round(t1);
        cx.attrs[(size_t)(f0 + wv) * C + k] =
          ext ? (int32_t)((o1 + kFpHalf) >> kFpFrac) : (int32_t)o1;
#if GPCC_FIN_VAR == 7   // every reconstruction store drained before the next source load
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        if (wv == 1) {
          const int64_t o0 = ext ? t0 : fp_round(t0);
          cx.attrs[(size_t)f0 * C + k] =
            ext ? (int32_t)((o0 + kFpHalf) >> kFpFrac) : (int32_t)o0;
        }
      }
    }
  }
#if GPCC_FIN_VAR == 2
  fin_check_lut(&lut_s, cx.lut, 1);
#endif
}

}  // namespace gpcc
