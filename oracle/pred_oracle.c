/* pred_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the PREDICTING transform of TMC13 given the LoD
 * structure (predictors in coding order), minus the entropy calls:
 *   computeQuantizationWeights            tmc3/PCCTMC3Common.h:895-922
 *   PCCPredictor::predictColor / predictReflectance
 *                                         tmc3/PCCTMC3Common.h:526-587
 *   predModeEligibleColor / ...Refl       tmc3/AttributeCommon.cpp:144-209
 *   encodeReflectancesPred, decidePredModeRefl, encodePredModeRefl,
 *   computeReflectanceResidual            tmc3/AttributeEncoder.cpp:647-853
 *   encodeColorsPred, decidePredModeColor, encodePredModeColor,
 *   computeColorResiduals, computeColorDistortions,
 *   computeInterComponentPredictionCoeffs tmc3/AttributeEncoder.cpp:857-1210,
 *                                         1652-1680
 *   PCCResidualsEncoder::bitsPt* / resStat* (the encoder's running rate
 *   model)                                tmc3/AttributeEncoder.cpp:127-222
 *   decodeReflectancesPred / decodePredModeRefl / decodeColorsPred /
 *   decodePredModeColor                   tmc3/AttributeDecoder.cpp:288-523
 * One loop over the predictors in coding order, as the reference runs it (the
 * encoder's mode decision reads a rate model that every earlier point has
 * updated: it is sequential by construction).  Pinned against the compiled
 * reference -- symbols recovered from its bitstream with its own entropy
 * decoder, and its reconstruction -- by tests/test_oracle_pred.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gpcc_attr_mi355.h"
#include "primitives.h"

/* computeQuantizationWeights (intra: no inter-frame references) */
void
oracle_pred_quant_weights(
  int32_t n, const int32_t* nc, const int32_t* ni, const int32_t* qnw, uint64_t* qw)
{
  for (int i = 0; i < n; i++)
    qw[i] = 1u << 8;
  for (int i = n - 1; i >= 0; i--)
    for (int j = 0; j < nc[i]; j++) {
      /* inter prediction: a neighbour in the reference frame (addressed behind the n
       * predictors, see pred_process) takes no share (PCCTMC3Common.h:913-914) */
      if (ni[3 * (size_t)i + j] >= n)
        continue;
      /* int32 * uint64 -> uint64: the reference lands in the UNSIGNED
       * overload of divExp2RoundHalfInf (PCCMath.h:678-685): modular product,
       * logical shift -- it matters once the weights wrap (shares that add up
       * to more than the weight itself on a deep structure) */
      qw[ni[3 * (size_t)i + j]] += ((uint64_t)(int64_t)qnw[j] * qw[i] + 128u) >> 8;
    }
}

static void
pred_quantizers(
  const gpcc_pred_params* p, int layer, const int32_t* qp_off, int point, quantizer_t q[2])
{
  const int o0 = qp_off ? qp_off[2 * (size_t)point] : 0, o1 = qp_off ? qp_off[2 * (size_t)point + 1] : 0;
  const int qp0 = clip_int(p->layer_qp[layer][0] + o0, 4, p->max_qp);
  const int qp1 = clip_int(p->layer_qp[layer][1] + o1 + qp0, 4, p->max_qp);
  q[0] = quantizer_make(qp0);
  q[1] = quantizer_make(qp1);
}

/* the running rate model of PCCResidualsEncoder */
typedef struct {
  int avail;
  int gt0[3], gt1[3];
} rate_model_t;

enum { kScaleRes = 1 << 20, kWindowLog2 = 6 };

static void
rate_update(rate_model_t* m, const int32_t* v, int c)
{
  for (int k = 0; k < c; k++) {
    m->gt0[k] += v[k] ? (kScaleRes - m->gt0[k]) >> kWindowLog2 : -(m->gt0[k] >> kWindowLog2);
    if (v[k])
      m->gt1[k] += abs(v[k]) > 1 ? (kScaleRes - m->gt1[k]) >> kWindowLog2 : -(m->gt1[k] >> kWindowLog2);
  }
}

static double
rate_bits_component(const rate_model_t* m, int k, int32_t value)
{
  const int l2 = 20; /* ilog2(scaleRes) */
  double bits = 0;
  bits += value ? l2 - log2(m->gt0[k]) : l2 - log2(kScaleRes - m->gt0[k]);
  const int mag = abs(value);
  if (mag) {
    bits += mag > 1 ? l2 - log2(m->gt1[k]) : l2 - log2(kScaleRes - m->gt1[k]);
    bits += 1;
    if (mag > 1)
      bits += 2.0 * log2(mag - 1.0) + 1.0;
  }
  return bits;
}

/* bitsPtRefl */
static double
rate_bits_refl(const rate_model_t* m, int32_t value, int mode)
{
  if (m->avail == 4) {
    value = (abs(value) << 2) + mode;
  } else if (m->avail == 3) {
    if (mode > 0)
      value = (abs(value) << 1) + (mode - 1);
    value = (abs(value) << 1) + (mode > 0);
  } else if (m->avail == 2) {
    value = (abs(value) << 1) + (mode & 1);
  }
  double bits = 0;
  bits += rate_bits_component(m, 0, value);
  return bits;
}

/* bitsPtColor */
static double
rate_bits_colour(const rate_model_t* m, const int64_t r[3], int mode)
{
  int32_t v[3] = {(int32_t)r[0], (int32_t)r[1], (int32_t)r[2]};
  if (m->avail == 4) {
    v[1] = 2 * abs(v[1]) + (mode >> 1);
    v[2] = 2 * abs(v[2]) + (mode & 1);
  } else if (m->avail == 3) {
    v[1] = 2 * abs(v[1]) + (mode > 0);
    if (mode > 0)
      v[2] = 2 * abs(v[2]) + (mode - 1);
  } else if (m->avail == 2) {
    v[1] = 2 * abs(v[1]) + (mode & 1);
  }
  double bits = 0;
  for (int k = 0; k < 3; k++)
    bits += rate_bits_component(m, k, v[k]);
  return bits;
}

/* PCCPredictor::predictColor / predictReflectance on the reconstruction
 * array rec [n][c] (coding order) */
static void
predict(
  int c, int mode, int cnt, const int32_t* ni, const int32_t* nw, const int32_t* rec,
  int64_t out[3])
{
  out[0] = out[1] = out[2] = 0;
  if (mode > cnt)
    return;
  if (mode > 0) {
    for (int k = 0; k < c; k++)
      out[k] = rec[(size_t)ni[mode - 1] * c + k];
    return;
  }
  for (int j = 0; j < cnt; j++)
    for (int k = 0; k < c; k++)
      out[k] += (int64_t)(uint32_t)nw[j] * rec[(size_t)ni[j] * c + k];
  for (int k = 0; k < c; k++)
    out[k] = (uint16_t)div_exp2_round_half_inf(out[k], 8);
}

static int
eligible(const gpcc_pred_params* p, int c, int cnt, const int32_t* ni, const int32_t* rec)
{
  if (cnt <= 1 || !p->max_num_direct_predictors)
    return 0;
  int64_t best = 0;
  for (int k = 0; k < c; k++) {
    int64_t lo = 0, hi = 0;
    for (int j = 0; j < cnt; j++) {
      const int64_t v = rec[(size_t)ni[j] * c + k];
      if (j == 0 || v < lo)
        lo = v;
      if (j == 0 || v > hi)
        hi = v;
    }
    if (k == 0 || hi - lo > best)
      best = hi - lo;
  }
  return best >= p->adaptive_prediction_threshold;
}

static int64_t
half_up8(int64_t x)
{
  return (x + 128) >> 8;
}

/* computeColorResiduals */
static void
colour_residuals(
  const gpcc_pred_params* p, const int64_t col[3], const int64_t pred[3], const int8_t icp[3],
  const quantizer_t q[2], int64_t r[3])
{
  r[0] = quantizer_quantize(q[0], (col[0] - pred[0]) << 8);
  const int64_t residual0 = half_up8(quantizer_scale(q[0], r[0]));
  for (int k = 1; k < 3; k++) {
    int64_t err = col[k] - pred[k];
    if (p->inter_component_prediction_enabled_flag)
      err -= (icp[k] * residual0 + 2) >> 2;
    r[k] = quantizer_quantize(q[1], err << 8);
  }
}

/* computeColorDistortions */
static int
colour_distortion(
  const gpcc_pred_params* p, const int64_t col[3], const int64_t pred[3], const quantizer_t q[2])
{
  const int64_t clip_max = ((int64_t)1 << p->bitdepth) - 1;
  int d = 0;
  for (int k = 0; k < 3; k++) {
    const quantizer_t qq = q[k ? 1 : 0];
    const int64_t rq = quantizer_quantize(qq, (col[k] - pred[k]) << 8);
    int64_t rec = pred[k] + half_up8(quantizer_scale(qq, rq));
    rec = rec < 0 ? 0 : (rec > clip_max ? clip_max : rec);
    d += abs((int)(col[k] - (int64_t)(uint16_t)rec));
  }
  return d;
}

/* computeInterComponentPredictionCoeffs */
static void
icp_coeffs(
  const gpcc_pred_params* p, int n, const int32_t* nc, const int32_t* ni, const int32_t* src,
  int8_t icp[GPCC_MAX_LODS][3])
{
  const int levels = p->max_num_detail_levels;
  for (int l = 0; l < GPCC_MAX_LODS; l++) {
    icp[l][0] = 0;
    icp[l][1] = icp[l][2] = l < levels ? 1 : 0;
  }
  int64_t sum_pred[8][3];
  int64_t sum_orig[3] = {0, 0, 0};
  memset(sum_pred, 0, sizeof(sum_pred));
  int lod = 0;
  for (int i = 0; i < n; i++) {
    /* predMode = 1: the first neighbour (nothing when there is none) */
    int32_t resid[3];
    for (int k = 0; k < 3; k++) {
      const int32_t pr = nc[i] >= 1 ? src[(size_t)ni[3 * (size_t)i] * 3 + k] : 0;
      resid[k] = src[(size_t)i * 3 + k] - pr;
    }
    for (int w = 0; w < 8; w++)
      for (int k = 1; k < 3; k++)
        sum_pred[w][k] += abs(resid[k] - icp[lod][k] * (((w + 1) * resid[0] + 2) >> 2));
    for (int k = 1; k < 3; k++)
      sum_orig[k] += abs(resid[k]);
    if (i != p->num_points_in_lod[lod] - 1)
      continue;
    for (int k = 1; k < 3; k++) {
      int best = 0;
      for (int w = 1; w < 8; w++)
        if (sum_pred[w][k] < sum_pred[best][k])
          best = w;
      icp[lod][k] = (int8_t)(icp[lod][k] * (1 + best));
      if (sum_pred[best][k] > sum_orig[k])
        icp[lod][k] = 0;
    }
    memset(sum_pred, 0, sizeof(sum_pred));
    sum_orig[1] = sum_orig[2] = 0;
    lod++;
  }
  for (; lod < GPCC_MAX_LODS; lod++)
    icp[lod][0] = icp[lod][1] = icp[lod][2] = 0;
}

static int
pred_process(
  int encoder, const gpcc_pred_params* p, int n, int c, const int32_t* nc, const int32_t* ni,
  const int32_t* nw, const int32_t* indexes, const int32_t* qp_off, int32_t* attrs,
  int32_t* values, int8_t* icp_io, int32_t* modes,
  /* attribute inter prediction (NULL: none): inter_ref [n][3] marks neighbours that live in
   * the reference frame (ni is then a point index there), attrs_ref [n_ref] its reflectances */
  const int32_t* inter_ref, const int32_t* attrs_ref, int n_ref)
{
  if (c != 1 && c != 3)
    return -1;
  /* Every place the reference reads a neighbour's value (predictReflectance
   * PCCTMC3Common.h:555-585, predModeEligibleRefl AttributeCommon.cpp:176-210,
   * decidePredModeRefl AttributeEncoder.cpp:663-717) takes the reference frame's reflectance
   * for such a neighbour: here the frame's values are appended to the reconstruction array
   * and the neighbour index points there. */
  int32_t* ni_ext = NULL;
  if (inter_ref) {
    if (c != 1 || p->scalable_lifting_enabled_flag)
      return -2; /* the reference has inter prediction in the reflectance driver only */
    ni_ext = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)n);
    for (size_t t = 0; t < 3 * (size_t)n; t++)
      ni_ext[t] = inter_ref[t] && (int)(t % 3) < nc[t / 3] ? n + ni[t] : ni[t];
    ni = ni_ext;
  } else
    n_ref = 0;
  const int maxcand = p->max_num_direct_predictors + !p->direct_avg_predictor_disabled_flag;
  const int dis = p->direct_avg_predictor_disabled_flag != 0;
  const int64_t clip_max = ((int64_t)1 << p->bitdepth) - 1;
  uint64_t* qw = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  int32_t* rec = (int32_t*)calloc(((size_t)n + n_ref) * c, sizeof(int32_t));  /* coding order */
  for (int r = 0; r < n_ref; r++)
    rec[(size_t)n + r] = attrs_ref[r];
  int32_t* src = (int32_t*)calloc((size_t)n * c, sizeof(int32_t));
  if (p->scalable_lifting_enabled_flag) {
    /* computeQuantizationWeightsScalable (PCCTMC3Common.h:858-891), whole slices */
    for (int i = 0; i < n; i++)
      qw[i] = 1u << 8;
    for (int l = 0; l < p->num_lods; l++) {
      const int start = l ? p->num_points_in_lod[l - 1] : 0;
      const uint64_t w = (uint64_t)(n / p->num_points_in_lod[l]) << 8;
      for (int i = start; i < p->num_points_in_lod[l]; i++)
        qw[i] = l == p->num_lods - 1 ? 1u << 8 : w;
    }
  } else
    oracle_pred_quant_weights(n, nc, ni, p->quant_neigh_weight, qw);
  if (encoder)
    for (int i = 0; i < n; i++)
      for (int k = 0; k < c; k++)
        src[(size_t)i * c + k] = attrs[(size_t)indexes[i] * c + k];

  int8_t icp[GPCC_MAX_LODS][3];
  memset(icp, 0, sizeof(icp));
  const int icp_present = c == 3 && p->inter_component_prediction_enabled_flag;
  if (icp_present) {
    if (encoder) {
      icp_coeffs(p, n, nc, ni, src, icp);
      memcpy(icp_io, icp, sizeof(icp));
    } else {
      memcpy(icp, icp_io, sizeof(icp));
    }
  }
  rate_model_t rm;
  rm.avail = maxcand;
  for (int k = 0; k < 3; k++)
    rm.gt0[k] = rm.gt1[k] = kScaleRes >> 1;

  int quant_layer = 0, lod = 0;
  const int8_t zero3[3] = {0, 0, 0};
  const int8_t* icpc = icp_present ? icp[0] : zero3;
  for (int i = 0; i < n; i++) {
    if (i == p->num_points_in_lod[quant_layer])
      quant_layer = quant_layer + 1 < p->num_qp_layers ? quant_layer + 1 : p->num_qp_layers - 1;
    if (icp_present && i == p->num_points_in_lod[lod])
      icpc = icp[++lod];
    quantizer_t q[2];
    pred_quantizers(p, quant_layer, qp_off, indexes[i], q);
    const int cnt = nc[i];
    const int32_t* pni = &ni[3 * (size_t)i];
    const int32_t* pnw = &nw[3 * (size_t)i];
    const int elig = eligible(p, c, cnt, pni, rec);
    int mode = 0;
    int64_t pr[3];
    int32_t val[3] = {0, 0, 0};
    int64_t col[3] = {0, 0, 0};
    if (encoder)
      for (int k = 0; k < c; k++)
        col[k] = src[(size_t)i * c + k];

    if (!encoder) {
      for (int k = 0; k < c; k++)
        val[k] = values[(size_t)i * c + k];
      if (elig) {
        if (c == 1) {
          /* decodePredModeRefl */
          int a = abs(val[0]);
          const int sg = val[0] < 0 ? -1 : 1;
          switch (maxcand) {
          case 4:
            mode = a & 3;
            val[0] = sg * (a >> 2);
            break;
          case 3:
            mode = a & 1;
            a >>= 1;
            if (mode > 0) {
              mode += a & 1;
              a >>= 1;
            }
            val[0] = sg * a;
            break;
          case 2:
            mode = a & 1;
            val[0] = sg * (a >> 1);
            break;
          default: mode = 0;
          }
        } else {
          /* decodePredModeColor */
          const int s1 = val[1] < 0 ? -1 : 1, s2 = val[2] < 0 ? -1 : 1;
          const int a1 = abs(val[1]), a2 = abs(val[2]);
          switch (maxcand) {
          case 4:
            val[1] = s1 * (a1 >> 1);
            val[2] = s2 * (a2 >> 1);
            mode = ((a1 & 1) << 1) + (a2 & 1);
            break;
          case 3:
            val[1] = s1 * (a1 >> 1);
            mode = a1 & 1;
            if (a1 & 1) {
              val[2] = s2 * (a2 >> 1);
              mode += a2 & 1;
            }
            break;
          case 2:
            val[1] = s1 * (a1 >> 1);
            mode = a1 & 1;
            break;
          default: mode = 0;
          }
        }
        mode += dis;
      }
    } else if (elig) {
      /* decidePredModeRefl / decidePredModeColor */
      mode = dis;
      predict(c, mode, cnt, pni, pnw, rec, pr);
      if (c == 1) {
        int64_t rq = quantizer_quantize(q[0], (col[0] - (int64_t)(uint64_t)pr[0]) << 8);
        int64_t best = (int64_t)rate_bits_refl(&rm, (int32_t)rq, mode - dis);
        for (int j = dis; j < cnt; j++) {
          if (j == p->max_num_direct_predictors)
            break;
          const int64_t np = rec[(size_t)pni[j]];
          rq = quantizer_quantize(q[0], (col[0] - np) << 8);
          const int64_t score = (int64_t)rate_bits_refl(&rm, (int32_t)rq, j + !dis);
          if (score < best) {
            best = score;
            mode = j + 1;
          }
        }
      } else {
        int64_t r[3];
        colour_residuals(p, col, pr, icpc, q, r);
        int dist = colour_distortion(p, col, pr, q);
        double rate = rate_bits_colour(&rm, r, 0);
        double best = dist + rate * 0.14 * (q[0].step >> 8);
        for (int j = dis; j < cnt; j++) {
          if (j == p->max_num_direct_predictors)
            break;
          int64_t np[3];
          for (int k = 0; k < 3; k++)
            np[k] = rec[(size_t)pni[j] * 3 + k];
          colour_residuals(p, col, np, icpc, q, r);
          dist = colour_distortion(p, col, np, q);
          rate = rate_bits_colour(&rm, r, j + !dis);
          const double score = dist + rate * 0.14 * (q[0].step >> 8);
          if (score < best) {
            best = score;
            mode = j + 1;
          }
        }
      }
    }
    if (modes)
      modes[i] = elig ? mode : -1;

    predict(c, mode, cnt, pni, pnw, rec, pr);
    int64_t residual0 = 0;
    for (int k = 0; k < c; k++) {
      const quantizer_t qq = q[k ? 1 : 0];
      int64_t weight = (int64_t)qw[i] < (int64_t)qq.step ? (int64_t)qw[i] : (int64_t)qq.step;
      weight >>= 8;
      int64_t rr;
      const int64_t icpterm = c == 3 ? (icpc[k] * residual0 + 2) >> 2 : 0;
      if (encoder) {
        int64_t residual = col[k] - pr[k];
        int64_t rq = quantizer_quantize(qq, (residual * weight) << 8);
        rr = half_up8(quantizer_scale(qq, rq)) / weight;
        if (c == 3 && p->inter_component_prediction_enabled_flag && k > 0) {
          residual -= icpterm;
          rq = quantizer_quantize(qq, (residual * weight) << 8);
          rr = half_up8(quantizer_scale(qq, rq)) / weight;
          rr += icpterm;
        }
        val[k] = (int32_t)rq;
      } else {
        rr = half_up8(quantizer_scale(qq, val[k])) / weight;
        /* the decoder adds the term unconditionally (icpCoeff is zero
         * when the tool is off) */
        const int64_t residual = rr;
        rr += icpterm;
        if (!k && p->inter_component_prediction_enabled_flag)
          residual0 = residual;
      }
      if (encoder && k == 0)
        residual0 = rr;
      int64_t v = pr[k] + rr;
      v = v < 0 ? 0 : (v > clip_max ? clip_max : v);
      rec[(size_t)i * c + k] = (int32_t)(uint16_t)v;
    }
    if (encoder) {
      if (elig) {
        const int m = mode - dis;
        if (c == 1) {
          /* encodePredModeRefl */
          const int sg = val[0] < 0 ? -1 : 1;
          int a = abs(val[0]);
          switch (maxcand) {
          case 4: val[0] = sg * ((a << 2) + m); break;
          case 3:
            if (m > 0)
              a = (a << 1) + (m - 1);
            a = (a << 1) + (m > 0);
            val[0] = sg * a;
            break;
          case 2: val[0] = sg * ((a << 1) + m); break;
          default: break;
          }
        } else {
          /* encodePredModeColor */
          const int s1 = val[1] < 0 ? -1 : 1, s2 = val[2] < 0 ? -1 : 1;
          const int a1 = abs(val[1]), a2 = abs(val[2]);
          switch (maxcand) {
          case 4:
            val[1] = s1 * ((a1 << 1) + (m >> 1));
            val[2] = s2 * ((a2 << 1) + (m & 1));
            break;
          case 3: {
            const int p1 = m ? 1 : 0;
            val[1] = s1 * ((a1 << 1) + p1);
            if (p1)
              val[2] = s2 * ((a2 << 1) + (m - p1));
            break;
          }
          case 2: val[1] = s1 * ((a1 << 1) + m); break;
          default: break;
          }
        }
      }
      rate_update(&rm, val, c);
      for (int k = 0; k < c; k++)
        values[(size_t)i * c + k] = val[k];
    }
  }
  for (int i = 0; i < n; i++)
    for (int k = 0; k < c; k++)
      attrs[(size_t)indexes[i] * c + k] = rec[(size_t)i * c + k];
  free(qw);
  free(rec);
  free(ni_ext);
  free(src);
  return 0;
}

int
oracle_pred_forward(
  const gpcc_pred_params* p, int32_t n, int32_t c, const int32_t* nc, const int32_t* ni,
  const int32_t* nw, const int32_t* indexes, const int32_t* qp_off, int32_t* attrs,
  int32_t* values, int8_t* icp, int32_t* modes)
{
  return pred_process(1, p, n, c, nc, ni, nw, indexes, qp_off, attrs, values, icp, modes, NULL, NULL, 0);
}

int
oracle_pred_inverse(
  const gpcc_pred_params* p, int32_t n, int32_t c, const int32_t* nc, const int32_t* ni,
  const int32_t* nw, const int32_t* indexes, const int32_t* qp_off, int32_t* attrs,
  int32_t* values, int8_t* icp, int32_t* modes)
{
  return pred_process(0, p, n, c, nc, ni, nw, indexes, qp_off, attrs, values, icp, modes, NULL, NULL, 0);
}

/* The reflectance predicting transform with attribute inter prediction
 * (encodeReflectancesPred / decodeReflectancesPred with enableAttrInterPred). */
int
oracle_pred_forward_inter(
  const gpcc_pred_params* p, int32_t n, const int32_t* nc, const int32_t* ni, const int32_t* nw,
  const int32_t* inter_ref, const int32_t* indexes, int32_t* attrs, const int32_t* attrs_ref,
  int32_t n_ref, int32_t* values, int32_t* modes)
{
  return pred_process(1, p, n, 1, nc, ni, nw, indexes, NULL, attrs, values, NULL, modes, inter_ref, attrs_ref, n_ref);
}

int
oracle_pred_inverse_inter(
  const gpcc_pred_params* p, int32_t n, const int32_t* nc, const int32_t* ni, const int32_t* nw,
  const int32_t* inter_ref, const int32_t* indexes, int32_t* attrs, const int32_t* attrs_ref,
  int32_t n_ref, int32_t* values, int32_t* modes)
{
  return pred_process(0, p, n, 1, nc, ni, nw, indexes, NULL, attrs, values, NULL, modes, inter_ref, attrs_ref, n_ref);
}
