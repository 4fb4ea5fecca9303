// ref_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Thin extern "C" wrapper that is compiled TOGETHER WITH the unmodified
// reference sources where they lie under /root/reference (see
// oracle/Makefile) into oracle/_ref/libtmc3_ref.so.  It only marshals the
// flattened parameter block of include/gpcc_attr_mi355.h into the
// reference's own structs and calls the reference's own functions:
//   pcc::regionAdaptiveHierarchicalTransform         tmc3/RAHT.cpp:1997
//   pcc::regionAdaptiveHierarchicalInverseTransform  tmc3/RAHT.cpp:2037
//   pcc::mortonAddr                                  tmc3/PCCMath.h:606
//   pcc::isqrt / pcc::irsqrt                         tmc3/misc.cpp:139/191
//   pcc::Quantizer / pcc::QpSet::quantizers          tmc3/quantization.{h,cpp}
//   pcc::FixedPoint                                  tmc3/FixedPoint.h
//   pcc::divApprox / pcc::ilog2 / pcc::morton3dAdd   tmc3/PCCMath.h, PCCMisc.h
// No reference source text is copied here.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may load the result.

#include <cstdint>
#include <cstring>
#include <vector>

#include "RAHT.h"
#include "FixedPoint.h"
#include "PCCMath.h"
#include "PCCMisc.h"
#include "quantization.h"
#include "hls.h"

#include "gpcc_attr_mi355.h"

namespace {

void
unflatten(
  const gpcc_raht_params& p, pcc::RahtPredictionParams* rp, pcc::QpSet* qs)
{
  rp->raht_prediction_enabled_flag = p.raht_prediction_enabled_flag != 0;
  rp->integer_haar_enable_flag = p.integer_haar_enable_flag != 0;
  rp->raht_prediction_threshold0 = p.raht_prediction_threshold0;
  rp->raht_prediction_threshold1 = p.raht_prediction_threshold1;
  rp->raht_subnode_prediction_enabled_flag =
    p.raht_subnode_prediction_enabled_flag != 0;
  rp->raht_prediction_search_range = p.raht_prediction_search_range;
  rp->predWeightParent.assign(
    p.pred_weight_parent, p.pred_weight_parent + 19);
  rp->predWeightChild.assign(p.pred_weight_child, p.pred_weight_child + 12);

  qs->layers.clear();
  for (int i = 0; i < p.num_qp_layers; i++)
    qs->layers.push_back({p.layer_qp[i][0], p.layer_qp[i][1]});
  qs->regions.clear();
  qs->rahtAcCoeffQps.clear();
  for (int i = 0; i < p.num_ac_qp_layers; i++) {
    std::vector<pcc::Qps> layer;
    for (int j = 0; j < 7; j++)
      layer.push_back({p.ac_qp_offset[i][j][0], p.ac_qp_offset[i][j][1]});
    qs->rahtAcCoeffQps.push_back(layer);
  }
  qs->maxQp = p.max_qp;
  qs->fixedPointQpOffset = p.fixed_point_qp_offset;
}

int
run(
  bool fwd,
  const gpcc_raht_params* p,
  const int64_t* morton,
  const int32_t* qp_off,
  int32_t* attrs,
  int32_t* coeffs,
  int n,
  int c)
{
  if (!p || !morton || !attrs || !coeffs || n <= 0 || c < 1 || c > 3)
    return -1;
  pcc::RahtPredictionParams rp;
  pcc::QpSet qs;
  unflatten(*p, &rp, &qs);

  std::vector<pcc::Qps> qps(n, pcc::Qps{0, 0});
  if (qp_off)
    for (int i = 0; i < n; i++)
      qps[i] = {qp_off[2 * i], qp_off[2 * i + 1]};

  // the reference takes a non-const pointer
  std::vector<int64_t> mc(morton, morton + n);

  pcc::AttributeInterPredParams inter;
  inter.enableAttrInterPred = false;
  inter.attrInterIntraSliceRDO = false;
  inter.frameDistance = 1;

  if (fwd)
    pcc::regionAdaptiveHierarchicalTransform(
      rp, qs, qps.data(), mc.data(), attrs, c, n, coeffs,
      p->raht_extension != 0, inter);
  else
    pcc::regionAdaptiveHierarchicalInverseTransform(
      rp, qs, qps.data(), mc.data(), attrs, c, n, coeffs,
      p->raht_extension != 0, inter);
  return 0;
}

}  // namespace

extern "C" {

int
ref_raht_forward(
  const gpcc_raht_params* p,
  const int64_t* morton,
  const int32_t* qp_off,
  int32_t* attrs,
  int32_t* coeffs,
  int32_t n,
  int32_t c)
{
  return run(true, p, morton, qp_off, attrs, coeffs, n, c);
}

int
ref_raht_inverse(
  const gpcc_raht_params* p,
  const int64_t* morton,
  const int32_t* qp_off,
  int32_t* attrs,
  int32_t* coeffs,
  int32_t n,
  int32_t c)
{
  return run(false, p, morton, qp_off, attrs, coeffs, n, c);
}

// RAHT with attribute inter prediction (AttributeInterPredParams::enableAttrInterPred,
// RAHT.cpp:1025-1345, 1540): morton_ref / attrs_ref [n_ref][c] the reference frame in Morton
// order.  tools: depth_minus1 (raht_inter_prediction_depth_minus1), layer_rdo
// (raht_enable_inter_intra_layer_RDO), filter_est (enableFilterEstimation), skip_layers
// (skipInitLayersForFiltering).  layer_modes [<= 32] / filter_taps [<= 32]: written by the
// encoder (counts in *num_modes / *num_taps), read by the decoder.
int
ref_raht_inter_qp(
  const gpcc_raht_params* p, int32_t fwd, const int64_t* morton, int32_t* attrs, int32_t* coeffs,
  int32_t n, int32_t c, const int64_t* morton_ref, const int32_t* attrs_ref, int32_t n_ref,
  int32_t depth_minus1, int32_t layer_rdo, int32_t filter_est, int32_t skip_layers,
  int32_t* layer_modes, int32_t* num_modes, int32_t* filter_taps, int32_t* num_taps, const int32_t* qp_off)
{
  if (!p || !morton || !attrs || !coeffs || n <= 0 || c < 1 || c > 3 || n_ref <= 0)
    return -1;
  pcc::RahtPredictionParams rp;
  pcc::QpSet qs;
  unflatten(*p, &rp, &qs);
  std::vector<pcc::Qps> qps(n, pcc::Qps{0, 0});
  if (qp_off)  // (region QP offsets per point, QpSet::regionQpOffset)
    for (int i = 0; i < n; i++)
      qps[i] = pcc::Qps{qp_off[2 * i], qp_off[2 * i + 1]};
  std::vector<int64_t> mc(morton, morton + n);
  pcc::AttributeInterPredParams inter;
  inter.enableAttrInterPred = true;
  inter.attrInterIntraSliceRDO = false;
  inter.frameDistance = 1;
  auto& ir = inter.paramsForInterRAHT;
  ir.voxelCount = n_ref;
  ir.mortonCode.assign(morton_ref, morton_ref + n_ref);
  ir.attributes.assign(attrs_ref, attrs_ref + size_t(n_ref) * c);
  ir.raht_inter_prediction_depth_minus1 = depth_minus1;
  ir.raht_inter_prediction_enabled = true;
  ir.raht_enable_inter_intra_layer_RDO = layer_rdo != 0;
  ir.enableFilterEstimation = filter_est != 0;
  ir.skipInitLayersForFiltering = skip_layers;
  if (!fwd) {
    inter.attr_layer_code_mode.assign(layer_modes, layer_modes + *num_modes);
    ir.FilterTaps.assign(filter_taps, filter_taps + *num_taps);
  }
  if (fwd)
    pcc::regionAdaptiveHierarchicalTransform(
      rp, qs, qps.data(), mc.data(), attrs, c, n, coeffs, p->raht_extension != 0, inter);
  else
    pcc::regionAdaptiveHierarchicalInverseTransform(
      rp, qs, qps.data(), mc.data(), attrs, c, n, coeffs, p->raht_extension != 0, inter);
  if (fwd) {
    *num_modes = std::min<int>(32, inter.attr_layer_code_mode.size());
    for (int i = 0; i < *num_modes; i++)
      layer_modes[i] = inter.attr_layer_code_mode[i];
    *num_taps = std::min<int>(32, ir.FilterTaps.size());
    for (int i = 0; i < *num_taps; i++)
      filter_taps[i] = ir.FilterTaps[i];
  }
  return 0;
}

int
ref_raht_inter(
  const gpcc_raht_params* p, int32_t fwd, const int64_t* morton, int32_t* attrs, int32_t* coeffs,
  int32_t n, int32_t c, const int64_t* morton_ref, const int32_t* attrs_ref, int32_t n_ref,
  int32_t depth_minus1, int32_t layer_rdo, int32_t filter_est, int32_t skip_layers,
  int32_t* layer_modes, int32_t* num_modes, int32_t* filter_taps, int32_t* num_taps)
{
  return ref_raht_inter_qp(
    p, fwd, morton, attrs, coeffs, n, c, morton_ref, attrs_ref, n_ref, depth_minus1, layer_rdo, filter_est, skip_layers,
    layer_modes, num_modes, filter_taps, num_taps, nullptr);
}

// The Morton prologue of encodeColorsTransformRaht
// (AttributeEncoder.cpp:1316-1321) on a raw xyz array.
int
ref_attr_morton_sort(
  const int32_t* xyz, int32_t n, int64_t* morton, int32_t* order)
{
  std::vector<pcc::MortonCodeWithIndex> packed(n);
  for (int i = 0; i < n; i++) {
    packed[i].mortonCode =
      pcc::mortonAddr(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    packed[i].index = i;
  }
  std::sort(packed.begin(), packed.end());
  for (int i = 0; i < n; i++) {
    morton[i] = packed[i].mortonCode;
    order[i] = packed[i].index;
  }
  return 0;
}

int64_t
ref_morton_addr(int32_t x, int32_t y, int32_t z)
{
  return pcc::mortonAddr(x, y, z);
}

uint64_t
ref_morton3d_add(uint64_t a, uint64_t b)
{
  return pcc::morton3dAdd(a, b);
}

uint32_t
ref_isqrt(uint64_t x)
{
  return pcc::isqrt(x);
}

uint64_t
ref_irsqrt(uint64_t x)
{
  return pcc::irsqrt(x);
}

int
ref_ilog2_u32(uint32_t x)
{
  return pcc::ilog2(x);
}

int
ref_ilog2_u64(uint64_t x)
{
  return pcc::ilog2(x);
}

int64_t
ref_fixedpoint_mul(int64_t a, int64_t b)
{
  pcc::FixedPoint fa, fb;
  fa.val = a;
  fb.val = b;
  fa *= fb;
  return fa.val;
}

int64_t
ref_fixedpoint_round(int64_t a)
{
  pcc::FixedPoint fa;
  fa.val = a;
  return fa.round();
}

int64_t
ref_fixedpoint_from_int(int64_t a)
{
  pcc::FixedPoint fa(a);
  return fa.val;
}

int64_t
ref_quantize(int32_t qp, int64_t x)
{
  return pcc::Quantizer(qp).quantize(x);
}

int64_t
ref_scale(int32_t qp, int64_t x)
{
  return pcc::Quantizer(qp).scale(x);
}

// step sizes of the two quantizers QpSet::quantizers(layer, off) selects
void
ref_qpset_steps(
  const gpcc_raht_params* p, int32_t layer, int32_t off0, int32_t off1,
  int32_t out_step[2])
{
  pcc::RahtPredictionParams rp;
  pcc::QpSet qs;
  unflatten(*p, &rp, &qs);
  auto q = qs.quantizers(layer, pcc::Qps{off0, off1});
  out_step[0] = q[0].stepSize();
  out_step[1] = q[1].stepSize();
}

int64_t
ref_div_exp2_round_half_up(int64_t x, int32_t s)
{
  return pcc::divExp2RoundHalfUp(x, s);
}

int64_t
ref_div_exp2_round_half_inf(int64_t x, int32_t s)
{
  return pcc::divExp2RoundHalfInf(x, s);
}

int64_t
ref_div_approx(int64_t a, uint64_t b, int32_t log2scale)
{
  return pcc::divApprox(a, b, log2scale);
}

}  // extern "C"

#include "io_hls.h"
#include "ref_lod_harness.inc"
