/* lift_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the lifting transform of TMC13 given the LoD
 * structure (predictors in coding order):
 *   PCCPredictor::computeWeights          tmc3/PCCTMC3Common.h:589-633
 *   PCCComputeQuantizationWeights         tmc3/PCCTMC3Common.h:828-854
 *   PCCLiftPredict / PCCLiftUpdate        tmc3/PCCTMC3Common.h:716-824
 *   encodeColorsLift / encodeReflectancesLift (minus the entropy calls)
 *                                         tmc3/AttributeEncoder.cpp:1379-1648
 *   computeLastComponentPredictionCoeff   tmc3/AttributeEncoder.cpp:1498-1539
 *   decodeColorsLift / decodeReflectancesLift (after the entropy decode)
 *                                         tmc3/AttributeDecoder.cpp:678-857
 * organised per level of detail the way the kernels are (every LoD is a
 * parallel step; the update is a scatter-add followed by one division per
 * receiving point).  Pinned against the compiled reference by
 * tests/test_oracle_lift.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gpcc_attr_mi355.h"
#include "primitives.h"

/* PCCPredictor::computeWeights: squared distances -> 8-bit weights */
void
oracle_compute_weights(int32_t n, int32_t* neigh_count, uint64_t* w)
{
  const uint64_t one = 1u << 8;
  for (int i = 0; i < n; i++) {
    uint64_t* d = &w[3 * i];
    int cnt = neigh_count[i];
    int sh = 0;
    while ((d[0] >> sh) >= one)
      sh++;
    if (sh > 0)
      for (int k = 0; k < cnt; k++)
        d[k] = (d[k] + ((uint64_t)1 << (sh - 1))) >> sh;
    while (cnt > 1 && d[cnt - 1] >= (d[0] << 8))
      cnt--;
    if (cnt <= 1) {
      d[0] = one;
    } else if (cnt == 2) {
      const uint64_t d0 = d[0], d1 = d[1];
      const uint64_t w1 = (uint64_t)div_approx((int64_t)d0, d0 + d1, 8);
      d[0] = (uint32_t)(one - w1);
      d[1] = (uint32_t)w1;
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
  }
}

static inline uint64_t
round_shift8_u64(uint64_t x)
{
  return (x + 128) >> 8;
}

/* PCCComputeQuantizationWeights */
/* inter prediction (xr != NULL): xr[3 i + j] marks a neighbour in the reference frame
 * (PCCNeighborInfo::interFrameRef); it takes no share (:845-846) */
static void
quant_weights(
  int n, const int32_t* nc, const int32_t* ni, const int32_t* nw, const int32_t* xr, uint64_t* qw)
{
  for (int i = 0; i < n; i++)
    qw[i] = 1u << 8;
  for (int i = n - 1; i >= 0; i--)
    for (int j = 0; j < nc[i]; j++) {
      if (xr && xr[3 * i + j])
        continue;
      qw[ni[3 * i + j]] += round_shift8_u64((uint64_t)(uint32_t)nw[3 * i + j] * qw[i]);
    }
}

/* computeQuantizationWeightsScalable (PCCTMC3Common.h:858-891) for a whole
 * slice (minGeomNodeSizeLog2 = 0): every predictor of a level of detail gets
 * numPoints / (points up to and including that level), the finest level 1 */
static void
quant_weights_scalable(int n, int lods, const int32_t* npl, uint64_t* qw)
{
  for (int i = 0; i < n; i++)
    qw[i] = 1u << 8;
  for (int l = 0; l < lods; l++) {
    const int start = l ? npl[l - 1] : 0;
    const uint64_t w = (uint64_t)(n / npl[l]) << 8;
    for (int i = start; i < npl[l]; i++)
      qw[i] = l == lods - 1 ? 1u << 8 : w;
  }
}

/* PCCLiftPredict over [start, end) */
static void
lift_predict(
  int c, int start, int end, int direct, const int32_t* nc, const int32_t* ni,
  const int32_t* nw, int64_t* a, const int32_t* xr, const int64_t* aref)
{
  for (int i = start; i < end; i++) {
    int64_t pred[3] = {0, 0, 0};
    for (int j = 0; j < nc[i]; j++)
      for (int k = 0; k < c; k++) {
        /* a neighbour of the reference frame contributes that frame's attribute, looked
         * up by its point index there (:735-740) */
        if (xr && xr[3 * i + j])
          pred[k] += (int64_t)(uint32_t)nw[3 * i + j] * aref[(size_t)ni[3 * i + j] * c + k];
        else
          pred[k] += (int64_t)(uint32_t)nw[3 * i + j] * a[(size_t)ni[3 * i + j] * c + k];
      }
    for (int k = 0; k < c; k++) {
      const int64_t p = div_exp2_round_half_inf(pred[k], 8);
      if (direct)
        a[(size_t)i * c + k] -= p;
      else
        a[(size_t)i * c + k] += p;
    }
  }
}

/* PCCLiftUpdate over [start, end): receivers are all points < start */
static void
lift_update(
  int c, int start, int end, int direct, const int32_t* nc, const int32_t* ni,
  const int32_t* nw, const uint64_t* qw, int64_t* a, uint64_t* uw, int64_t* up, const int32_t* xr)
{
  memset(uw, 0, sizeof(uint64_t) * (size_t)start);
  memset(up, 0, sizeof(int64_t) * (size_t)start * c);
  for (int i = start; i < end; i++)
    for (int j = 0; j < nc[i]; j++) {
      if (xr && xr[3 * i + j])
        continue; /* nothing flows back into the reference frame (:799-800) */
      const uint64_t wgt = round_shift8_u64((uint64_t)(uint32_t)nw[3 * i + j] * qw[i]);
      const int nb = ni[3 * i + j];
      uw[nb] += wgt;
      for (int k = 0; k < c; k++)
        up[(size_t)nb * c + k] =
          (int64_t)((uint64_t)up[(size_t)nb * c + k] + wgt * (uint64_t)a[(size_t)i * c + k]);
    }
  for (int i = 0; i < start; i++) {
    const uint32_t sum = (uint32_t)uw[i]; /* NB: truncated to 32 bits, :813 */
    if (!sum)
      continue;
    for (int k = 0; k < c; k++) {
      const int64_t u = div_approx(up[(size_t)i * c + k], sum, 0);
      if (direct)
        a[(size_t)i * c + k] += u;
      else
        a[(size_t)i * c + k] -= u;
    }
  }
}

static void
lift_quantizers(
  const gpcc_lift_params* p, int layer, const int32_t* qp_off, int point,
  quantizer_t q[2])
{
  const int o0 = qp_off ? qp_off[2 * point] : 0, o1 = qp_off ? qp_off[2 * point + 1] : 0;
  const int qp0 = clip_int(p->layer_qp[layer][0] + o0, 4, p->max_qp);
  const int qp1 = clip_int(p->layer_qp[layer][1] + o1 + qp0, 4, p->max_qp);
  q[0] = quantizer_make(qp0 + p->fixed_point_qp_offset);
  q[1] = quantizer_make(qp1 + p->fixed_point_qp_offset);
}

static int
lift_process(
  int encoder, const gpcc_lift_params* p, int n, int c, const int32_t* nc,
  const int32_t* ni, const int32_t* nw, const int32_t* indexes,
  const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int8_t* lcp,
  const int32_t* xr, const int32_t* attrs_ref, int n_ref)
{
  const int lods = p->num_lods;
  int64_t* aref = NULL;
  if (xr) {
    if (c != 1 || p->scalable_lifting_enabled_flag)
      return -2; /* the reference has inter prediction in the reflectance driver only */
    aref = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_ref > 0 ? n_ref : 1));
    for (int i = 0; i < n_ref; i++)
      aref[i] = (int64_t)attrs_ref[i] * 256;
  }
  const int32_t* npl = p->num_points_in_lod;
  uint64_t* qw = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  uint64_t* uw = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  int64_t* up = (int64_t*)malloc(sizeof(int64_t) * (size_t)n * c);
  int64_t* a = (int64_t*)calloc((size_t)n * c, sizeof(int64_t));
  if (p->scalable_lifting_enabled_flag)
    quant_weights_scalable(n, lods, npl, qw);
  else
    quant_weights(n, nc, ni, nw, xr, qw);

  int8_t signs[GPCC_MAX_LODS];
  memset(signs, 0, sizeof(signs));
  if (encoder) {
    for (int i = 0; i < n; i++)
      for (int k = 0; k < c; k++)
        a[(size_t)i * c + k] = (int64_t)attrs[(size_t)indexes[i] * c + k] * 256;
    for (int l = lods - 1; l >= 1; l--) {
      lift_predict(c, npl[l - 1], npl[l], 1, nc, ni, nw, a, xr, aref);
      lift_update(c, npl[l - 1], npl[l], 1, nc, ni, nw, qw, a, uw, up, xr);
    }
    if (c == 3 && p->last_component_prediction_enabled_flag) {
      int64_t s12 = 0, s11 = 0;
      int lod = 0;
      for (int i = 0; i < n; i++) {
        /* NB: the products are truncated to int (:1510-1511) */
        s12 += (int32_t)(a[3 * (size_t)i + 1] * a[3 * (size_t)i + 2]);
        s11 += (int32_t)(a[3 * (size_t)i + 1] * a[3 * (size_t)i + 1]);
        if (lod >= lods || i != npl[lod] - 1)
          continue;
        int scale = 0;
        if (s12 && s11) {
          const int sign = ((s12 < 0) ^ (s11 < 0)) ? -1 : 1;
          scale = (int)(((s12 << 2) + sign * (s11 >> 1)) / s11);
        }
        s12 = s11 = 0;
        signs[lod++] = (int8_t)clip_int(scale, -8, 8);
      }
      for (; lod < GPCC_MAX_LODS; lod++)
        signs[lod] = lod ? signs[lod - 1] : 0;
      memcpy(lcp, signs, sizeof(signs));
    }
  } else if (c == 3 && p->last_component_prediction_enabled_flag) {
    memcpy(signs, lcp, sizeof(signs));
  }

  /* quantise / de-quantise every coefficient (coding order) */
  int quant_layer = 0, lod = 0;
  int lcpc = signs[0];
  for (int i = 0; i < n; i++) {
    if (i == npl[quant_layer])
      quant_layer = quant_layer + 1 < p->num_qp_layers ? quant_layer + 1 : p->num_qp_layers - 1;
    if (lod < lods && i == npl[lod]) {
      lod++;
      lcpc = signs[lod < GPCC_MAX_LODS ? lod : GPCC_MAX_LODS - 1];
    }
    quantizer_t q[2];
    lift_quantizers(p, quant_layer, qp_off, indexes[i], q);
    const int64_t iqw = (int64_t)irsqrt_u64(qw[i]);
    const int64_t qwt = (int64_t)((qw[i] * (uint64_t)iqw + ((uint64_t)1 << 39)) >> 40);
    int64_t* col = &a[(size_t)i * c];
    int32_t* val = &coeffs[(size_t)i * c];
    if (c == 1) {
      if (encoder)
        val[0] = (int32_t)quantizer_quantize(q[0], col[0] * qwt);
      col[0] = div_exp2_round_half_inf(quantizer_scale(q[0], val[0]) * iqw, 40);
      continue;
    }
    /* colour (c == 3; c == 2 follows the same chain without component 2) */
    if (encoder)
      val[0] = (int32_t)quantizer_quantize(q[0], col[0] * qwt);
    int64_t scaled = quantizer_scale(q[0], val[0]);
    col[0] = div_exp2_round_half_inf(scaled * iqw, 40);
    if (encoder)
      val[1] = (int32_t)quantizer_quantize(q[1], col[1] * qwt);
    scaled = quantizer_scale(q[1], val[1]);
    col[1] = div_exp2_round_half_inf(scaled * iqw, 40);
    if (c == 3) {
      if (encoder)
        col[2] -= (lcpc * col[1]) >> 2;
      scaled *= lcpc;
      scaled >>= 2;
      if (encoder)
        val[2] = (int32_t)quantizer_quantize(q[1], col[2] * qwt);
      scaled += quantizer_scale(q[1], val[2]);
      col[2] = div_exp2_round_half_inf(scaled * iqw, 40);
    }
  }

  /* inverse lifting */
  for (int l = 1; l < lods; l++) {
    lift_update(c, npl[l - 1], npl[l], 0, nc, ni, nw, qw, a, uw, up, xr);
    lift_predict(c, npl[l - 1], npl[l], 0, nc, ni, nw, a, xr, aref);
  }
  const int64_t clip_max = ((int64_t)1 << p->bitdepth) - 1;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < c; k++) {
      int64_t v = div_exp2_round_half_inf(a[(size_t)i * c + k], 8);
      v = v < 0 ? 0 : (v > clip_max ? clip_max : v);
      attrs[(size_t)indexes[i] * c + k] = (int32_t)v;
    }
  free(qw);
  free(uw);
  free(up);
  free(a);
  free(aref);
  return 0;
}

int
oracle_lift_forward(
  const gpcc_lift_params* p, int32_t n, int32_t c, const int32_t* nc,
  const int32_t* ni, const int32_t* nw, const int32_t* indexes,
  const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int8_t* lcp)
{
  return lift_process(1, p, n, c, nc, ni, nw, indexes, qp_off, attrs, coeffs, lcp, NULL, NULL, 0);
}

int
oracle_lift_inverse(
  const gpcc_lift_params* p, int32_t n, int32_t c, const int32_t* nc,
  const int32_t* ni, const int32_t* nw, const int32_t* indexes,
  const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int8_t* lcp)
{
  return lift_process(0, p, n, c, nc, ni, nw, indexes, qp_off, attrs, coeffs, lcp, NULL, NULL, 0);
}

/* Reflectance lifting with attribute inter prediction (encodeReflectancesLift /
 * decodeReflectancesLift with AttributeInterPredParams::enableAttrInterPred,
 * AttributeEncoder.cpp:1543-1648, AttributeDecoder.cpp:780-857): inter_ref [n][3] marks the
 * neighbours that live in the reference frame (neigh_index is then a point index there),
 * attrs_ref [n_ref] that frame's reflectances. */
int
oracle_lift_forward_inter(
  const gpcc_lift_params* p, int32_t n, const int32_t* nc, const int32_t* ni, const int32_t* nw,
  const int32_t* inter_ref, const int32_t* indexes, int32_t* attrs, const int32_t* attrs_ref,
  int32_t n_ref, int32_t* coeffs)
{
  return lift_process(1, p, n, 1, nc, ni, nw, indexes, NULL, attrs, coeffs, NULL, inter_ref, attrs_ref, n_ref);
}

int
oracle_lift_inverse_inter(
  const gpcc_lift_params* p, int32_t n, const int32_t* nc, const int32_t* ni, const int32_t* nw,
  const int32_t* inter_ref, const int32_t* indexes, int32_t* attrs, const int32_t* attrs_ref,
  int32_t n_ref, int32_t* coeffs)
{
  return lift_process(0, p, n, 1, nc, ni, nw, indexes, NULL, attrs, coeffs, NULL, inter_ref, attrs_ref, n_ref);
}
