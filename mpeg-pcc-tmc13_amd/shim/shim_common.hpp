// shim_common.hpp -- what the replacement translation units share: the device
// context of the process, the fall-back policy, the flattening of the
// reference's parameter sets into the POD blocks of include/gpcc_attr_mi355.h,
// and the residual syntax (zero runs, coefficient tuples) written / parsed
// through the reference's PUBLIC entropy interface (tmc3/entropy.h:
// EntropyEncoder / EntropyDecoder / AdaptiveBitModel) over the context models
// of AttributeContexts (tmc3/AttributeCommon.h:47-58).
//
// Built against the reference's headers; contains no reference code.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "AttributeCommon.h"
#include "PCCTMC3Common.h"
#include "entropy.h"
#include "hls.h"
#include "quantization.h"

#include "gpcc_attr_mi355.h"

namespace gpcc_shim {

// one context per process (the reference is single threaded), made on first use
inline gpcc_ctx*
process_context(const char* what)
{
  static gpcc_ctx* ctx = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // the library must have been built from THIS header: the parameter blocks are passed by pointer and their
    // sizes changed between ABI versions (a shorter block would be read past its end)
    if (gpcc_abi_version() != GPCC_ABI_VERSION) {
      std::fprintf(
        stderr, "gpcc: libgpcc_attr_mi355 has ABI %d, this binary was built against %d; %s stays on the CPU\n",
        gpcc_abi_version(), GPCC_ABI_VERSION, what);
      return nullptr;
    }
    const char* dev = std::getenv("GPCC_DEVICE");
    if (gpcc_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &ctx) != GPCC_OK) {
      std::fprintf(stderr, "gpcc: no MI355X context (%s); %s stays on the CPU\n", gpcc_last_error(), what);
      ctx = nullptr;
    }
    // GPCC_RESERVE_POINTS=n (the sequence's largest slice): reserve once, before the first slice (gpcc_ctx_reserve)
    const char* rsv = std::getenv("GPCC_RESERVE_POINTS");
    if (ctx && rsv && std::atoll(rsv) > 0 && gpcc_ctx_reserve(ctx, std::atoll(rsv), 1, 3) != GPCC_OK)
      std::fprintf(stderr, "gpcc: GPCC_RESERVE_POINTS=%s not reserved (%s); workspace grows on demand\n", rsv, gpcc_last_error());
  }
  return ctx;
}

// GPCC_STRICT=1: work that cannot run on the device is an error, not a CPU
// fallback (CI on a GPU box)
inline void
strict_check(const char* what)
{
  const char* strict = std::getenv("GPCC_STRICT");
  if (strict && strict[0] == '1') {
    std::fprintf(stderr, "gpcc: GPCC_STRICT=1 and %s did not run on the device (%s)\n", what, gpcc_last_error());
    std::abort();
  }
}

// The LoD fields of the APS / ABH (buildPredictorsFast's inputs, hls.h:782-876).
// false: the block cannot express these parameters -> CPU path
inline bool
flatten_lod(
  const pcc::AttributeParameterSet& aps, const pcc::AttributeBrickHeader& abh,
  int minGeomNodeSizeLog2, const pcc::AttributeInterPredParams& inter,
  gpcc_lod_params* lp, bool inter_allowed = false)
{
  // (attribute inter prediction: only the caller that has an entry for it says so --
  // AttributeLods::generate -> gpcc_lod_build_inter)
  if ((inter.enableAttrInterPred && !inter_allowed) || minGeomNodeSizeLog2 > 0)
    return false;
  if (aps.num_detail_levels_minus1 + 1 >= GPCC_MAX_LODS)
    return false;
  *lp = gpcc_lod_params{};
  lp->attr_encoding = int(aps.attr_encoding);
  lp->lod_decimation_type = int(aps.lod_decimation_type);
  lp->num_detail_levels_minus1 = aps.num_detail_levels_minus1;
  lp->num_pred_nearest_neighbours_minus1 = aps.num_pred_nearest_neighbours_minus1;
  lp->intra_lod_search_range = aps.intra_lod_search_range;
  lp->inter_lod_search_range = aps.inter_lod_search_range;
  lp->prediction_with_distribution_enabled = aps.predictionWithDistributionEnabled;
  for (int k = 0; k < 3; k++)
    lp->lod_neigh_bias[k] = aps.lodNeighBias[k];
  lp->intra_lod_prediction_skip_layers = aps.intra_lod_prediction_skip_layers;
  lp->dist2 = aps.dist2;
  lp->attr_dist2_delta = abh.attr_dist2_delta;
  lp->canonical_point_order_flag = aps.canonical_point_order_flag;
  lp->max_points_per_sort_log2_plus1 = aps.max_points_per_sort_log2_plus1;
  lp->scalable_lifting_enabled_flag = aps.scalable_lifting_enabled_flag;
  lp->max_neigh_range_minus1 = aps.max_neigh_range_minus1;
  lp->pred_weight_blending_enabled_flag =
    aps.attr_encoding == pcc::AttributeEncoding::kPredictingTransform
    && aps.pred_weight_blending_enabled_flag;
  for (size_t i = 0; i < aps.lodSamplingPeriod.size() && i < GPCC_MAX_LODS; i++)
    lp->lod_sampling_period[i] = aps.lodSamplingPeriod[i];
  return true;
}

// The LoD structure an attribute coder object of the reference CACHES: AttributeEncoder / AttributeDecoder
// generate `_lods` for the first LoD-based attribute they code and every later attribute coded by the same
// object runs over that structure (AttributeEncoder.cpp:484-490, AttributeDecoder.cpp:217-221:
// `if (aps.lodParametersPresent() && _lods.empty()) _lods.generate(...)`), whatever its own parameter set
// says; the caller replaces the object when AttributeLods::isReusable (AttributeCommon.cpp:76-139) says
// no -- a test that does NOT compare attr_encoding (weight blending and the search inside a LoD belong to
// the predicting transform), predictionWithDistributionEnabled or the inter-prediction state.  The device
// entries build the structure with every call, so the factories' objects remember under WHICH parameters
// the reference's cached structure was built and build with those.
struct FirstLods {
  bool have = false;
  pcc::AttributeParameterSet aps;
  pcc::AttributeBrickHeader abh;
  bool inter = false;

  void note(
    const pcc::AttributeParameterSet& a, const pcc::AttributeBrickHeader& b,
    const pcc::AttributeInterPredParams& ip)
  {
    if (have || !a.lodParametersPresent())
      return;
    have = true;
    aps = a;
    abh = b;
    inter = ip.enableAttrInterPred;
  }

  // AttributeLods::isReusable over what `note` recorded: the interface contract of
  // AttributeEncoderIntf::isReusable (Attribute.h:101-103), field by field as the reference compares them
  bool reusable(const pcc::AttributeParameterSet& a, const pcc::AttributeBrickHeader& b) const
  {
    if (!have || !a.lodParametersPresent())
      return true;
    if (aps.scalable_lifting_enabled_flag || a.scalable_lifting_enabled_flag)
      return false;
    return aps.num_pred_nearest_neighbours_minus1 == a.num_pred_nearest_neighbours_minus1
      && aps.inter_lod_search_range == a.inter_lod_search_range
      && aps.intra_lod_search_range == a.intra_lod_search_range
      && aps.num_detail_levels_minus1 == a.num_detail_levels_minus1 && aps.lodNeighBias == a.lodNeighBias
      && aps.lod_decimation_type == a.lod_decimation_type
      && aps.dist2 + abh.attr_dist2_delta == a.dist2 + b.attr_dist2_delta
      && aps.lodSamplingPeriod == a.lodSamplingPeriod
      && aps.intra_lod_prediction_skip_layers == a.intra_lod_prediction_skip_layers
      && aps.canonical_point_order_flag == a.canonical_point_order_flag
      && aps.max_points_per_sort_log2_plus1 == a.max_points_per_sort_log2_plus1
      && aps.pred_weight_blending_enabled_flag == a.pred_weight_blending_enabled_flag;
  }
};

// While a factory object hands a slice to the reference's coder, the structure that coder generates must
// be the one its cache WOULD hold: seam 2 (AttributeLods_mi355.cpp) builds with these parameter sets
// instead of the ones it is called with.  (One instance per shared object; the reference is single threaded.)
struct LodOverride {
  const pcc::AttributeParameterSet* aps = nullptr;
  const pcc::AttributeBrickHeader* abh = nullptr;
};
inline LodOverride&
lod_override()
{
  static LodOverride o;
  return o;
}
struct ScopedLodOverride {
  explicit ScopedLodOverride(const FirstLods& f)
  {
    if (f.have) {
      lod_override().aps = &f.aps;
      lod_override().abh = &f.abh;
    }
  }
  ~ScopedLodOverride() { lod_override() = LodOverride{}; }
};

// RahtPredictionParams + QpSet (hls.h, quantization.h:124-139) -> gpcc_raht_params, for an intra slice.
// false: the block cannot express these parameters (or the slice uses attribute inter prediction,
// which is not on the device for RAHT) -> CPU path
inline bool
flatten_raht(
  const pcc::RahtPredictionParams& rp, const pcc::QpSet& qs, bool extension,
  const pcc::AttributeInterPredParams& inter, gpcc_raht_params* p)
{
  if (inter.enableAttrInterPred)
    return false;
  if (rp.predWeightParent.size() != 19)
    return false;
  if (rp.raht_subnode_prediction_enabled_flag && rp.predWeightChild.size() != 12)
    return false;
  if (qs.layers.empty() || qs.layers.size() > GPCC_MAX_QP_LAYERS)
    return false;
  if (qs.rahtAcCoeffQps.size() > GPCC_MAX_AC_QP_LAYERS)
    return false;
  *p = gpcc_raht_params{};
  p->raht_prediction_enabled_flag = rp.raht_prediction_enabled_flag;
  p->integer_haar_enable_flag = rp.integer_haar_enable_flag;
  p->raht_prediction_threshold0 = rp.raht_prediction_threshold0;
  p->raht_prediction_threshold1 = rp.raht_prediction_threshold1;
  p->raht_subnode_prediction_enabled_flag = rp.raht_subnode_prediction_enabled_flag;
  p->raht_prediction_search_range = rp.raht_prediction_search_range;
  for (int i = 0; i < 19; i++)
    p->pred_weight_parent[i] = rp.predWeightParent[i];
  for (size_t i = 0; i < 12 && i < rp.predWeightChild.size(); i++)
    p->pred_weight_child[i] = rp.predWeightChild[i];
  p->raht_extension = extension;
  p->num_qp_layers = int(qs.layers.size());
  for (size_t i = 0; i < qs.layers.size(); i++) {
    p->layer_qp[i][0] = qs.layers[i][0];
    p->layer_qp[i][1] = qs.layers[i][1];
  }
  p->max_qp = qs.maxQp;
  p->fixed_point_qp_offset = qs.fixedPointQpOffset;
  p->num_ac_qp_layers = int(qs.rahtAcCoeffQps.size());
  for (size_t i = 0; i < qs.rahtAcCoeffQps.size(); i++) {
    if (qs.rahtAcCoeffQps[i].size() != 7)
      return false;
    for (int j = 0; j < 7; j++) {
      p->ac_qp_offset[i][j][0] = qs.rahtAcCoeffQps[i][j][0];
      p->ac_qp_offset[i][j][1] = qs.rahtAcCoeffQps[i][j][1];
    }
  }
  return true;
}

// QpSet (quantization.h:124-139) -> the layer table and the QP regions of gpcc_lift_params /
// gpcc_pred_params.  false: more layers or regions than the block holds
template<class Params>
inline bool
flatten_qp(const pcc::QpSet& qpSet, Params* p)
{
  if (qpSet.layers.empty() || qpSet.layers.size() > GPCC_MAX_QP_LAYERS || qpSet.regions.size() > GPCC_MAX_QP_REGIONS)
    return false;
  p->num_qp_regions = int(qpSet.regions.size());
  for (int r = 0; r < p->num_qp_regions; r++) {
    for (int k = 0; k < 3; k++) {
      p->qp_region_min[r][k] = qpSet.regions[r].region.min[k];
      p->qp_region_max[r][k] = qpSet.regions[r].region.max[k];
    }
    p->qp_region_offset[r][0] = qpSet.regions[r].qpOffset[0];
    p->qp_region_offset[r][1] = qpSet.regions[r].qpOffset[1];
  }
  p->num_qp_layers = int(qpSet.layers.size());
  for (int l = 0; l < p->num_qp_layers; l++) {
    p->layer_qp[l][0] = qpSet.layers[l][0];
    p->layer_qp[l][1] = qpSet.layers[l][1];
  }
  p->max_qp = qpSet.maxQp;
  return true;
}

// the slice's QP regions for the RAHT slice drivers (gpcc_qp_regions).  false: more regions than the block holds
inline bool
flatten_regions(const pcc::QpSet& qpSet, gpcc_qp_regions* p)
{
  if (qpSet.regions.size() > GPCC_MAX_QP_REGIONS)
    return false;
  *p = gpcc_qp_regions{};
  p->num_qp_regions = int(qpSet.regions.size());
  for (int r = 0; r < p->num_qp_regions; r++) {
    for (int k = 0; k < 3; k++) {
      p->qp_region_min[r][k] = qpSet.regions[r].region.min[k];
      p->qp_region_max[r][k] = qpSet.regions[r].region.max[k];
    }
    p->qp_region_offset[r][0] = qpSet.regions[r].qpOffset[0];
    p->qp_region_offset[r][1] = qpSet.regions[r].qpOffset[1];
  }
  return true;
}

inline void
positions_of(const pcc::PCCPointSet3& cloud, std::vector<int32_t>* xyz)
{
  const size_t n = cloud.getPointCount();
  xyz->resize(3 * n);
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++)
      (*xyz)[3 * i + k] = cloud[i][k];
}

inline void
attributes_of(const pcc::PCCPointSet3& cloud, int c, std::vector<int32_t>* a)
{
  const size_t n = cloud.getPointCount();
  a->resize(size_t(c) * n);
  for (size_t i = 0; i < n; i++) {
    if (c == 3) {
      const auto col = cloud.getColor(i);
      for (int k = 0; k < 3; k++)
        (*a)[3 * i + k] = col[k];
    } else
      (*a)[i] = cloud.getReflectance(i);
  }
}

inline void
store_attributes(const std::vector<int32_t>& a, int c, pcc::PCCPointSet3* cloud)
{
  const size_t n = cloud->getPointCount();
  for (size_t i = 0; i < n; i++) {
    if (c == 3)
      cloud->setColor(
        i, pcc::Vec3<pcc::attr_t>{pcc::attr_t(a[3 * i]), pcc::attr_t(a[3 * i + 1]), pcc::attr_t(a[3 * i + 2])});
    else
      cloud->setReflectance(i, pcc::attr_t(a[i]));
  }
}

// A slice with attribute inter prediction (one component: the reference's reflectance
// drivers): the LoD structure with neighbours in the reference frame
// (gpcc_lod_build_inter) and what the transforms over it need
struct InterStructure {
  std::vector<int32_t> nc, ni, nw, idx, xr, attrsFrame;
  int32_t npl[GPCC_MAX_LODS];
  int32_t nl = 0;
  int nFrame = 0;
};

inline int
build_inter_structure(
  gpcc_ctx* ctx, const gpcc_lod_params& lod, const std::vector<int32_t>& xyz, int n,
  const pcc::AttributeBrickHeader& abh, const pcc::AttributeInterPredParams& inter, InterStructure* s)
{
  const auto& frame = inter.referencePointCloud;
  s->nFrame = int(frame.getPointCount());
  if (s->nFrame <= 0 || !frame.hasReflectances()) {
    std::fprintf(stderr, "gpcc: the reference frame of this slice has no reflectances; it stays on the CPU\n");
    return GPCC_ERR_UNSUPPORTED;
  }
  std::vector<int32_t> xyzFrame;
  positions_of(frame, &xyzFrame);
  s->attrsFrame.resize(s->nFrame);
  for (int i = 0; i < s->nFrame; i++)
    s->attrsFrame[i] = frame.getReflectance(i);
  s->nc.resize(n);
  s->ni.resize(size_t(n) * 3);
  s->nw.resize(size_t(n) * 3);
  s->idx.resize(n);
  s->xr.resize(size_t(n) * 3);
  return gpcc_lod_build_inter(
    ctx, &lod, xyz.data(), n, xyzFrame.data(), s->nFrame, abh.attrInterPredSearchRange, inter.frameDistance,
    s->nc.data(), s->ni.data(), s->nw.data(), s->idx.data(), s->npl, &s->nl, s->xr.data());
}

// The context models of an attribute slice by the ids gpcc_binarise_symbols
// writes: 0..4 ctxRunLen, 5..18 ctxCoeffGtN[2][7], 19..24 ctxCoeffRemPrefix[2][3],
// 25..30 ctxCoeffRemSuffix[2][3] (the declaration order of AttributeContexts)
struct SliceContexts : pcc::AttributeContexts {
  explicit SliceContexts(const pcc::AttributeContexts& saved) : pcc::AttributeContexts(saved) {}
  const pcc::AttributeContexts& saved() const { return *this; }

  pcc::AdaptiveBitModel& model(int id)
  {
    if (id < 5)
      return ctxRunLen[id];
    if (id < 19)
      return ctxCoeffGtN[(id - 5) / 7][(id - 5) % 7];
    if (id < 25)
      return ctxCoeffRemPrefix[(id - 19) / 3][(id - 19) % 3];
    return ctxCoeffRemSuffix[(id - 25) / 3][(id - 25) % 3];
  }

  // ---- the residual syntax, parsing side (7.3.4 of the G-PCC text: zero_run_length,
  //      the coefficient tuple with its cross-component context selection, signs;
  //      the encoder's side of it is gpcc_binarise_symbols) -------------------------
  int parse_run_length(pcc::EntropyDecoder& ac)
  {
    int id = 0, run = 0;
    // unary part, a context per position: up to three
    while (run < 3) {
      if (!ac.decode(model(id)))
        return run;
      run++, id++;
    }
    // pairs under one context: up to four, the pair's low bit bypassed
    for (int pairs = 0; pairs < 4; pairs++) {
      if (!ac.decode(model(3)))
        return run + int(ac.decode());
      run += 2;
    }
    return run + int(ac.decodeExpGolomb(2, model(4)));
  }

  // magnitude: > 0, > 1, then an exp-Golomb remainder (k = 1)
  int parse_magnitude(pcc::EntropyDecoder& ac, int gt0, int gt1, int rem)
  {
    if (!ac.decode(ctxCoeffGtN[0][gt0]))
      return 0;
    if (!ac.decode(ctxCoeffGtN[1][gt1]))
      return 1;
    return 2 + int(ac.decodeExpGolomb(1, ctxCoeffRemPrefix[rem], ctxCoeffRemSuffix[rem]));
  }

  // three components: the second is coded first, then the third, then the first,
  // each conditioned on "== 0" / "<= 1" of the ones before; an all-zero tuple
  // cannot occur (it would have been part of the run), so the first component is
  // coded minus one when the other two are zero
  void parse_tuple(pcc::EntropyDecoder& ac, int32_t v[3])
  {
    const int m1 = parse_magnitude(ac, 0, 0, 1);
    const int z1 = m1 == 0, s1 = m1 <= 1;
    const int m2 = parse_magnitude(ac, 1 + z1, 1 + s1, 1);
    const int z2 = m2 == 0, s2 = m2 <= 1;
    int m0 = parse_magnitude(ac, 3 + 2 * z1 + z2, 3 + 2 * s1 + s2, 0);
    m0 += z1 & z2;
    v[0] = (m0 && ac.decode()) ? -m0 : m0;
    v[1] = (m1 && ac.decode()) ? -m1 : m1;
    v[2] = (m2 && ac.decode()) ? -m2 : m2;
  }

  // one component: never zero outside a run, so magnitude minus one, then the sign
  int32_t parse_scalar(pcc::EntropyDecoder& ac)
  {
    const int m = parse_magnitude(ac, 0, 0, 0) + 1;
    return ac.decode() ? -m : m;
  }

  // all n predictors of a slice, zero runs expanded: values [n][c] coding order
  void parse_slice(pcc::EntropyDecoder& ac, int n, int c, int32_t* values)
  {
    int left = 0;  // zeros still to come before the next coded position
    for (int i = 0; i < n; i++) {
      if (--left < 0)
        left = parse_run_length(ac);
      int32_t* v = values + size_t(c) * i;
      if (left) {
        for (int k = 0; k < c; k++)
          v[k] = 0;
      } else if (c == 3)
        parse_tuple(ac, v);
      else
        v[0] = parse_scalar(ac);
    }
  }
};

}  // namespace gpcc_shim
