// AttributeLods_mi355.cpp -- drop-in replacement for the reference's
//   pcc::AttributeLods::generate     (tmc3/AttributeCommon.cpp:44-72,
//                                     declared tmc3/AttributeCommon.h:80-87)
// the single producer of the LoD structure (predictors, numPointsInLod,
// indexes) every lifting / predicting attribute coder consumes
// (AttributeEncoder.cpp:575-579, AttributeDecoder.cpp:292-296).
//
// `generate` is a member function, so the link seam is made with one macro:
// the integrator compiles tmc3/AttributeCommon.cpp and the two-line adapter
// AttributeLods_cpu_adapter.cpp with -Dgenerate=generateCpu (the reference
// body then defines AttributeLods::generateCpu, isReusable etc. are
// untouched) and adds this translation unit, which defines
// AttributeLods::generate with the original signature.  The LoD fields of
// the APS / ABH are flattened into gpcc_lod_params, the build runs on the
// MI355X through the C ABI and the public vectors are filled; whenever the
// device path declines (no GPU, inter prediction, scalable lifting, ...) the renamed reference body runs instead.
//
// Built against the reference's headers; contains no reference code.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "AttributeCommon.h"
#include "PCCTMC3Common.h"

#include "gpcc_attr_mi355.h"

namespace gpcc_shim {
// AttributeLods_cpu_adapter.cpp: calls the renamed reference member
void lods_generate_cpu(
  pcc::AttributeLods& lods, const pcc::AttributeParameterSet& aps,
  const pcc::AttributeBrickHeader& abh, int geom_num_points_minus1,
  int minGeomNodeSizeLog2, const pcc::PCCPointSet3& cloud,
  const pcc::AttributeInterPredParams& attrInterPredParams);

// what this TU did (gpcc_shim_lod_counters)
long long g_lod_device_calls = 0, g_lod_cpu_calls = 0;

gpcc_ctx*
lod_device_context()
{
  static gpcc_ctx* ctx = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* dev = std::getenv("GPCC_DEVICE");
    if (gpcc_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &ctx) != GPCC_OK) {
      std::fprintf(
        stderr, "gpcc: no MI355X context (%s); LoD build stays on the CPU\n",
        gpcc_last_error());
      ctx = nullptr;
    }
  }
  return ctx;
}
}  // namespace gpcc_shim

namespace pcc {

namespace {

// false: the block cannot express these parameters -> CPU path
bool
flatten(
  const AttributeParameterSet& aps, const AttributeBrickHeader& abh,
  int minGeomNodeSizeLog2, const AttributeInterPredParams& inter,
  gpcc_lod_params* lp)
{
  if (inter.enableAttrInterPred || minGeomNodeSizeLog2 > 0)
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
    aps.attr_encoding == AttributeEncoding::kPredictingTransform
    && aps.pred_weight_blending_enabled_flag;
  for (size_t i = 0; i < aps.lodSamplingPeriod.size() && i < GPCC_MAX_LODS; i++)
    lp->lod_sampling_period[i] = aps.lodSamplingPeriod[i];
  return true;
}

}  // namespace

void
AttributeLods::generate(
  const AttributeParameterSet& aps, const AttributeBrickHeader& abh,
  int geom_num_points_minus1, int minGeomNodeSizeLog2,
  const PCCPointSet3& cloud, const AttributeInterPredParams& attrInterPredParams)
{
  gpcc_lod_params lp;
  gpcc_ctx* ctx = gpcc_shim::lod_device_context();
  const int n = int(cloud.getPointCount());
  if (ctx && n > 0 && flatten(aps, abh, minGeomNodeSizeLog2, attrInterPredParams, &lp)) {
    static_assert(sizeof(point_t) == 3 * sizeof(int32_t), "Vec3<int32_t> is three ints");
    std::vector<int32_t> xyz(size_t(n) * 3);
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++)
        xyz[3 * size_t(i) + k] = cloud[i][k];
    std::vector<int32_t> nc(n), ni(size_t(n) * 3), nw(size_t(n) * 3), idx(n);
    int32_t npl[GPCC_MAX_LODS], nl = 0;
    int rc = gpcc_lod_build(
      ctx, &lp, xyz.data(), n, nc.data(), ni.data(), nw.data(), idx.data(), npl, &nl);
    if (rc == GPCC_OK) {
      _aps = aps;
      _abh = abh;
      predictors.assign(n, PCCPredictor());
      for (int i = 0; i < n; i++) {
        auto& p = predictors[i];
        p.init();
        p.predMode = 0;
        p.neighborCount = nc[i];
        for (int k = 0; k < 3; k++) {
          p.neighbors[k].predictorIndex = ni[3 * size_t(i) + k];
          p.neighbors[k].weight = nw[3 * size_t(i) + k];
          p.neighbors[k].pointIndex = k < nc[i] ? idx[ni[3 * size_t(i) + k]] : 0;
          p.neighbors[k].interFrameRef = false;
        }
      }
      indexes.assign(idx.begin(), idx.end());
      numPointsInLod.assign(npl, npl + nl);
      gpcc_shim::g_lod_device_calls++;
      return;
    }
    if (rc != GPCC_ERR_UNSUPPORTED)
      std::fprintf(stderr, "gpcc: %s; LoD build falls back to the CPU\n", gpcc_last_error());
  }
  gpcc_shim::g_lod_cpu_calls++;
  {
    const char* strict = std::getenv("GPCC_STRICT");
    if (strict && strict[0] == '1') {
      std::fprintf(stderr, "gpcc: GPCC_STRICT=1 and AttributeLods::generate did not run on the device (%s)\n", gpcc_last_error());
      std::abort();
    }
  }
  gpcc_shim::lods_generate_cpu(
    *this, aps, abh, geom_num_points_minus1, minGeomNodeSizeLog2, cloud,
    attrInterPredParams);
}

}  // namespace pcc

// {LoD builds that ran on the device, builds handed to the reference's CPU body}
extern "C" void
gpcc_shim_lod_counters(long long out[2])
{
  out[0] = gpcc_shim::g_lod_device_calls;
  out[1] = gpcc_shim::g_lod_cpu_calls;
}
