// lift_kernels.hpp -- the lifting transform on gfx950, given the LoD
// structure (predictors in coding order).
//
// The reference walks the predictors one by one (PCCLiftPredict /
// PCCLiftUpdate / PCCComputeQuantizationWeights, tmc3/PCCTMC3Common.h:716-854)
// but a point only ever reads / updates points of COARSER levels of detail
// (assert at :743, :805), so a level of detail is one parallel step:
//   predict : gather, one thread per point of the LoD
//   update  : scatter-add of (weight, weight * value) into the receivers
//             with 64-bit atomics -- integer adds commute, so the sums are
//             the reference's -- then one divApprox per receiver, which
//             also clears the accumulators for the next LoD
//   quant weights: the same scatter, finest LoD first.
// Quantisation (encode/decode{Colors,Reflectances}Lift,
// AttributeEncoder.cpp:1427-1473 / :1599-1623, AttributeDecoder.cpp:713-749 /
// :817-837) is one thread per coefficient; the arithmetic coder that the
// reference interleaves with it consumes the quantised values afterwards.
#pragma once

#include "raht_common.hpp"

namespace gpcc {

constexpr int kMaxLodRanges = GPCC_MAX_LODS + 1;

struct LiftCtx {
  int32_t n, c;
  int32_t num_lods;
  int32_t npl[GPCC_MAX_LODS];     // cumulative LoD sizes
  // state of the reference's running `quantLayer` / `lod` counters for the
  // points of each range between distinct LoD boundaries (built on the host
  // by replaying AttributeEncoder.cpp:1430-1440)
  int32_t num_ranges;
  int32_t range_start[kMaxLodRanges];
  int32_t range_qlayer[kMaxLodRanges];
  int32_t range_lcp[kMaxLodRanges];
  int32_t lcp_enabled;
  int32_t bitdepth;
  int32_t num_qp_layers;
  int32_t layer_qp[GPCC_MAX_QP_LAYERS][2];
  int32_t max_qp, fixed_point_qp_offset;
  const int32_t* nc;       // [n]
  const int32_t* ni;       // [n][3]
  const int32_t* nw;       // [n][3]
  const int32_t* indexes;  // [n]
  const int32_t* qp_off;   // [n][2] by point, or null
  int32_t* attrs;          // [n][c] point order
  int32_t* coeffs;         // [n][c] coding order
  int8_t* lcp;             // [GPCC_MAX_LODS]
  int64_t* a;              // [n][c] working values, coding order
  unsigned long long* qw;  // [n] quantisation weights
  unsigned long long* uw;  // [n] update weight sums
  unsigned long long* up;  // [n][c] update sums
  long long* lcp_sums;     // [GPCC_MAX_LODS][2]
  const RsqrtLut* rsqrt;   // device copy of the rsqrt tables
};

__device__ __forceinline__ int64_t
div_exp2_round_half_inf(int64_t x, int s)
{
  return round_shift_sym(x, s);  // sign-symmetric, PCCMath.h:665-673
}

// divApprox (tmc3/PCCMath.h:715-737); the 256-entry LUT is
// round(65536 / i) - 1 (checked against the reference symbol by the tests)
__device__ __forceinline__ int64_t
div_approx(int64_t a, uint64_t b, int log2scale)
{
  int nn = ilog2_u64(b) + 1 - 8;
  nn = nn < 0 ? 0 : nn;
  const uint32_t index = (uint32_t)((b + (((uint64_t)1 << nn) >> 1)) >> nn);
  const int64_t inv = (int64_t)((2u * 65536u + index) / (2u * index) - 1u) + 1;
  return (inv * a) >> (nn + 16 - log2scale);
}

// PCCPredictor::computeWeights (tmc3/PCCTMC3Common.h:589-633)
__global__ __launch_bounds__(256) void
lod_compute_weights_kernel(
  int n, int32_t* neigh_count, const uint64_t* __restrict__ dist2, int32_t* weight)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    uint64_t d[3] = {dist2[3 * (size_t)i], dist2[3 * (size_t)i + 1], dist2[3 * (size_t)i + 2]};
    int cnt = neigh_count[i];
    const uint64_t one = 256;
    // smallest n with (d0 >> n) < 256
    int sh = bitlen64(d[0]) - 8;
    sh = sh < 0 ? 0 : sh;
    if (sh > 0)
      for (int k = 0; k < 3; k++)
        if (k < cnt)
          d[k] = (d[k] + ((uint64_t)1 << (sh - 1))) >> sh;
    while (cnt > 1 && d[cnt - 1] >= (d[0] << 8))
      cnt--;
    if (cnt <= 1) {
      d[0] = one;
    } else if (cnt == 2) {
      const uint64_t w1 = (uint64_t)div_approx((int64_t)d[0], d[0] + d[1], 8);
      d[1] = (uint32_t)w1;
      d[0] = (uint32_t)(one - w1);
    } else {
      cnt = 3;
      const uint64_t d0 = d[0], d1 = d[1], d2 = d[2];
      const uint64_t sum = d1 * d2 + d0 * d2 + d0 * d1;
      const uint64_t w2 = (uint64_t)div_approx((int64_t)(d0 * d1), sum, 8);
      const uint64_t w1 = (uint64_t)div_approx((int64_t)(d0 * d2), sum, 8);
      d[0] = (uint32_t)(one - (w1 + w2));
      d[1] = (uint32_t)w1;
      d[2] = (uint32_t)w2;
    }
    neigh_count[i] = cnt;
    for (int k = 0; k < 3; k++)
      weight[3 * (size_t)i + k] = (int32_t)d[k];
  }
}

// PCCPredictor::blendWeights (tmc3/PCCTMC3Common.h:635-693; predicting
// transform only, AttributeCommon.cpp:66-69).  neigh_point holds the POINT
// indices of the neighbours, xyz the positions in point order.
__global__ __launch_bounds__(256) void
lod_blend_weights_kernel(
  int n, const int32_t* __restrict__ neigh_count, const int32_t* __restrict__ neigh_point,
  const int32_t* __restrict__ xyz, int32_t* weight)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    if (neigh_count[i] != 3)
      continue;
    const int32_t* q0 = &xyz[3 * (size_t)neigh_point[3 * (size_t)i]];
    const int32_t* q1 = &xyz[3 * (size_t)neigh_point[3 * (size_t)i + 1]];
    const int32_t* q2 = &xyz[3 * (size_t)neigh_point[3 * (size_t)i + 2]];
    int64_t d01 = 0, d02 = 0, d12 = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int64_t a = (int64_t)q0[c] - q1[c], b = (int64_t)q0[c] - q2[c], e = (int64_t)q1[c] - q2[c];
      d01 += a * a;
      d02 += b * b;
      d12 += e * e;
    }
    constexpr int dd = 10, bb = 1, cc = 5;
    const int b1 = d01 <= d02 ? bb : cc;
    const int b2 = d01 <= d12 ? cc : bb;
    const int b3 = d02 <= d12 ? bb : cc;
    const int w0 = weight[3 * (size_t)i], w1 = weight[3 * (size_t)i + 1], w2 = weight[3 * (size_t)i + 2];
    const int v0 = (w0 * dd + w1 * (16 - dd - b2) + w2 * b3) >> 4;
    const int v1 = (w0 * b1 + w1 * dd + w2 * (16 - dd - b3)) >> 4;
    weight[3 * (size_t)i] = v0;
    weight[3 * (size_t)i + 1] = v1;
    weight[3 * (size_t)i + 2] = 256 - v0 - v1;
  }
}

__global__ __launch_bounds__(256) void
lift_init_kernel(LiftCtx cx, int encoder)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cx.n;
       i += gridDim.x * blockDim.x) {
    cx.qw[i] = 256;
    cx.uw[i] = 0;
    for (int k = 0; k < cx.c; k++) {
      cx.up[(size_t)i * cx.c + k] = 0;
      cx.a[(size_t)i * cx.c + k] =
        encoder ? (int64_t)cx.attrs[(size_t)cx.indexes[i] * cx.c + k] * 256 : 0;
    }
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * GPCC_MAX_LODS;
       i += gridDim.x * blockDim.x)
    cx.lcp_sums[i] = 0;
}

// PCCComputeQuantizationWeights, one LoD [start, end)
__global__ __launch_bounds__(256) void
lift_quant_weights_kernel(LiftCtx cx, int start, int end)
{
  for (int i = start + blockIdx.x * blockDim.x + threadIdx.x; i < end;
       i += gridDim.x * blockDim.x) {
    const unsigned long long q = cx.qw[i];
    const int cnt = cx.nc[i];
    for (int j = 0; j < cnt; j++)
      atomicAdd(
        &cx.qw[cx.ni[3 * (size_t)i + j]],
        ((unsigned long long)(uint32_t)cx.nw[3 * (size_t)i + j] * q + 128) >> 8);
  }
}

// computeQuantizationWeightsScalable (PCCTMC3Common.h:858-891), whole slices:
// numPoints / (points up to and including the predictor's level), the finest
// level 1.  Shared by the lifting and the predicting transform.
struct LodSizes {
  int32_t num_lods;
  int32_t npl[GPCC_MAX_LODS];  // cumulative, coarse to fine
};

__global__ __launch_bounds__(256) void
quant_weights_scalable_kernel(int n, LodSizes t, unsigned long long* qw)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int l = 0;
    while (l < t.num_lods - 1 && t.npl[l] <= i)
      l++;
    qw[i] = l == t.num_lods - 1 ? 256ull : (unsigned long long)(n / t.npl[l]) << 8;
  }
}

template<int C>
__global__ __launch_bounds__(256) void
lift_predict_kernel(LiftCtx cx, int start, int end, int direct)
{
  for (int i = start + blockIdx.x * blockDim.x + threadIdx.x; i < end;
       i += gridDim.x * blockDim.x) {
    int64_t pred[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = 0;
    const int cnt = cx.nc[i];
    for (int j = 0; j < cnt; j++) {
      const int64_t w = (uint32_t)cx.nw[3 * (size_t)i + j];
      const size_t nb = cx.ni[3 * (size_t)i + j];
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] += w * cx.a[nb * C + k];
    }
#pragma unroll
    for (int k = 0; k < C; k++) {
      const int64_t p = div_exp2_round_half_inf(pred[k], 8);
      cx.a[(size_t)i * C + k] += direct ? -p : p;
    }
  }
}

template<int C>
__global__ __launch_bounds__(256) void
lift_update_scatter_kernel(LiftCtx cx, int start, int end)
{
  for (int i = start + blockIdx.x * blockDim.x + threadIdx.x; i < end;
       i += gridDim.x * blockDim.x) {
    const unsigned long long q = cx.qw[i];
    const int cnt = cx.nc[i];
    for (int j = 0; j < cnt; j++) {
      const unsigned long long w =
        ((unsigned long long)(uint32_t)cx.nw[3 * (size_t)i + j] * q + 128) >> 8;
      const size_t nb = cx.ni[3 * (size_t)i + j];
      atomicAdd(&cx.uw[nb], w);
#pragma unroll
      for (int k = 0; k < C; k++)
        atomicAdd(&cx.up[nb * C + k], w * (unsigned long long)cx.a[(size_t)i * C + k]);
    }
  }
}

template<int C>
__global__ __launch_bounds__(256) void
lift_update_apply_kernel(LiftCtx cx, int start, int direct)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < start;
       i += gridDim.x * blockDim.x) {
    const uint32_t sum = (uint32_t)cx.uw[i];  // truncated as in the reference (:813)
    if (cx.uw[i])
      cx.uw[i] = 0;
#pragma unroll
    for (int k = 0; k < C; k++) {
      const int64_t u = (int64_t)cx.up[(size_t)i * C + k];
      if (u)
        cx.up[(size_t)i * C + k] = 0;
      if (sum) {
        const int64_t v = div_approx(u, sum, 0);
        cx.a[(size_t)i * C + k] += direct ? v : -v;
      }
    }
  }
}

// per-LoD sums of computeLastComponentPredictionCoeff (:1498-1539)
__global__ __launch_bounds__(256) void
lift_lcp_sums_kernel(LiftCtx cx)
{
  const int lane = threadIdx.x & 63;
  // wave-uniform loop: every lane of a wavefront takes part in the shuffles
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < cx.n;
       base += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)base + lane;
    const bool valid = i < cx.n;
    long long m12 = 0, m11 = 0;
    int l = 0;
    if (valid) {
      const int64_t k1 = cx.a[(size_t)i * 3 + 1], k2 = cx.a[(size_t)i * 3 + 2];
      // NB: the reference truncates both products to int before summing
      m12 = (int32_t)(k1 * k2);
      m11 = (int32_t)(k1 * k1);
      while (l < cx.num_lods - 1 && i >= cx.npl[l])
        l++;
    }
    // one atomic pair per wavefront when its lanes share the LoD (almost
    // always: indices are LoD-ordered); integer sums, order irrelevant
    const int l0 = __shfl(l, 0);
    if (__all(!valid || l == l0)) {
      long long s12 = m12, s11 = m11;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        s12 += __shfl_xor(s12, d);
        s11 += __shfl_xor(s11, d);
      }
      if (lane == 0) {
        atomicAdd((unsigned long long*)&cx.lcp_sums[2 * l0], (unsigned long long)s12);
        atomicAdd((unsigned long long*)&cx.lcp_sums[2 * l0 + 1], (unsigned long long)s11);
      }
    } else if (valid) {
      atomicAdd((unsigned long long*)&cx.lcp_sums[2 * l], (unsigned long long)m12);
      atomicAdd((unsigned long long*)&cx.lcp_sums[2 * l + 1], (unsigned long long)m11);
    }
  }
}

__global__ void
lift_lcp_resolve_kernel(LiftCtx cx)
{
  if (blockIdx.x || threadIdx.x)
    return;
  // replay of the sequential loop over coefficients on per-LoD sums: the
  // sums accumulate until an index equals numPointsInLod[lod] - 1
  long long s12 = 0, s11 = 0;
  int lod = 0;
  int8_t signs[GPCC_MAX_LODS];
  for (int l = 0; l < GPCC_MAX_LODS; l++)
    signs[l] = 0;
  int prev_end = 0;
  for (int l = 0; l < cx.num_lods; l++) {
    const int end = cx.npl[l];
    if (end == prev_end)
      continue;  // empty LoD: no coefficient index falls here
    s12 += cx.lcp_sums[2 * l];
    s11 += cx.lcp_sums[2 * l + 1];
    prev_end = end;
    if (lod < cx.num_lods && end == cx.npl[lod]) {
      int scale = 0;
      if (s12 && s11) {
        const int sign = ((s12 < 0) ^ (s11 < 0)) ? -1 : 1;
        scale = (int)(((s12 << 2) + sign * (s11 >> 1)) / s11);
      }
      s12 = s11 = 0;
      signs[lod++] = (int8_t)clip(scale, -8, 8);
    }
  }
  for (; lod < GPCC_MAX_LODS; lod++)
    signs[lod] = lod ? signs[lod - 1] : 0;
  for (int l = 0; l < GPCC_MAX_LODS; l++)
    cx.lcp[l] = signs[l];
}

template<int C>
__global__ __launch_bounds__(256) void
lift_quantise_kernel(LiftCtx cx, int encoder)
{
  GPCC_VGPR_FLOOR_64();
  __shared__ RsqrtLut lut;
  for (int i = threadIdx.x; i < 96; i += blockDim.x) {
    lut.r3[i] = cx.rsqrt->r3[i];
    lut.rc[i] = cx.rsqrt->rc[i];
  }
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cx.n;
       i += gridDim.x * blockDim.x) {
    int r = 0;
    while (r + 1 < cx.num_ranges && i >= cx.range_start[r + 1])
      r++;
    const int layer = cx.range_qlayer[r];
    const int lcpc = (C == 3 && cx.lcp_enabled) ? cx.lcp[cx.range_lcp[r]] : 0;
    const int pt = cx.indexes[i];
    const int o0 = cx.qp_off ? cx.qp_off[2 * (size_t)pt] : 0;
    const int o1 = cx.qp_off ? cx.qp_off[2 * (size_t)pt + 1] : 0;
    const int qp0 = clip(cx.layer_qp[layer][0] + o0, 4, cx.max_qp);
    const int qp1 = clip(cx.layer_qp[layer][1] + o1 + qp0, 4, cx.max_qp);
    const Quantizer q0 = make_quantizer(qp0 + cx.fixed_point_qp_offset);
    const Quantizer q1 = make_quantizer(qp1 + cx.fixed_point_qp_offset);
    const uint64_t w = cx.qw[i];
    const int64_t iqw = (int64_t)irsqrt(w, lut);
    const int64_t qwt = (int64_t)((w * (uint64_t)iqw + ((uint64_t)1 << 39)) >> 40);
    int64_t* col = &cx.a[(size_t)i * C];
    int32_t* val = &cx.coeffs[(size_t)i * C];
    int32_t v0 = encoder ? (int32_t)quantize(q0, col[0] * qwt) : val[0];
    int64_t scaled = (int64_t)v0 * q0.step;
    col[0] = div_exp2_round_half_inf(scaled * iqw, 40);
    if (encoder)
      val[0] = v0;
    if (C >= 2) {
      int32_t v1 = encoder ? (int32_t)quantize(q1, col[1] * qwt) : val[1];
      scaled = (int64_t)v1 * q1.step;
      col[1] = div_exp2_round_half_inf(scaled * iqw, 40);
      if (encoder)
        val[1] = v1;
    }
    if (C == 3) {
      if (encoder)
        col[2] -= ((int64_t)lcpc * col[1]) >> 2;
      scaled *= lcpc;
      scaled >>= 2;
      int32_t v2 = encoder ? (int32_t)quantize(q1, col[2] * qwt) : val[2];
      scaled += (int64_t)v2 * q1.step;
      col[2] = div_exp2_round_half_inf(scaled * iqw, 40);
      if (encoder)
        val[2] = v2;
    }
  }
}

__global__ __launch_bounds__(256) void
lift_writeback_kernel(LiftCtx cx)
{
  const int64_t clip_max = ((int64_t)1 << cx.bitdepth) - 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cx.n;
       i += gridDim.x * blockDim.x) {
    const size_t pt = cx.indexes[i];
    for (int k = 0; k < cx.c; k++) {
      int64_t v = div_exp2_round_half_inf(cx.a[(size_t)i * cx.c + k], 8);
      v = v < 0 ? 0 : (v > clip_max ? clip_max : v);
      cx.attrs[pt * cx.c + k] = (int32_t)v;
    }
  }
}

// QP regions of a slice (the qp_region_* fields of gpcc_lift_params / gpcc_pred_params) -> the region
// offset of every POINT, as QpSet::regionQpOffset (tmc3/quantization.cpp:195-204) gives it: the first
// region that contains the point, bounds inclusive (Box3::contains, PCCMath.h:469-474)
struct QpRegionSet {
  int32_t n;
  int32_t lo[GPCC_MAX_QP_REGIONS][3], hi[GPCC_MAX_QP_REGIONS][3], off[GPCC_MAX_QP_REGIONS][2];
};

__global__ __launch_bounds__(256) void
qp_region_fill_kernel(const int32_t* __restrict__ xyz, int n, QpRegionSet rs, int32_t* __restrict__ qp_off)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
  int o0 = 0, o1 = 0;
  bool found = false;
  for (int r = 0; r < rs.n; r++) {
    const bool in = !(x < rs.lo[r][0] || x > rs.hi[r][0] || y < rs.lo[r][1] || y > rs.hi[r][1] || z < rs.lo[r][2]
                      || z > rs.hi[r][2]);
    if (in && !found) {
      found = true;
      o0 = rs.off[r][0];
      o1 = rs.off[r][1];
    }
  }
  qp_off[2 * (size_t)i] = o0;
  qp_off[2 * (size_t)i + 1] = o1;
}


}  // namespace gpcc
