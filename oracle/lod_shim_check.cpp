// lod_shim_check.cpp -- TEST INFRASTRUCTURE.  Links, exactly as
// INTEGRATION.md describes, the reference's AttributeCommon.cpp (its
// AttributeLods::generate renamed generateCpu), the adapter, the drop-in
// translation unit mpeg-pcc-tmc13_amd/shim/AttributeLods_mi355.cpp and the
// HIP library, then calls AttributeLods::generate THROUGH THE REFERENCE'S OWN
// C++ SIGNATURE on a seeded cloud and compares every predictor, the coding
// order and the LoD sizes with the renamed CPU implementation.
// Exit code 0 = identical.  Prints which path ran.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "AttributeCommon.h"
#include "PCCTMC3Common.h"
#include "gpcc_attr_mi355.h"

namespace gpcc_shim {
void lods_generate_cpu(
  pcc::AttributeLods&, const pcc::AttributeParameterSet&,
  const pcc::AttributeBrickHeader&, int, int, const pcc::PCCPointSet3&,
  const pcc::AttributeInterPredParams&);
}

static uint64_t
rng(uint64_t& s)
{
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  return s;
}

int
main(int argc, char** argv)
{
  const int n = argc > 1 ? std::atoi(argv[1]) : 40000;
  const int lifting = argc > 2 ? std::atoi(argv[2]) : 1;
  uint64_t s = 88172645463325252ull;
  // a noisy sheet z = f(x, y) in a 2^9 cube, unique voxels not required
  pcc::PCCPointSet3 cloud;
  cloud.resize(n);
  for (int i = 0; i < n; i++) {
    const int x = int(rng(s) % 512), y = int(rng(s) % 512);
    const int z = (x * x / 700 + y / 3 + int(rng(s) % 3)) % 512;
    cloud[i] = pcc::point_t{x, y, z};
  }

  pcc::AttributeParameterSet aps;
  aps.attr_encoding = lifting ? pcc::AttributeEncoding::kLiftingTransform
                              : pcc::AttributeEncoding::kPredictingTransform;
  aps.lod_decimation_type = pcc::LodDecimationMethod::kNone;
  aps.canonical_point_order_flag = false;
  aps.max_points_per_sort_log2_plus1 = 0;
  aps.num_pred_nearest_neighbours_minus1 = 2;
  aps.intra_lod_search_range = lifting ? 0 : 16;
  aps.inter_lod_search_range = 1100000;
  aps.predictionWithDistributionEnabled = true;
  aps.lodNeighBias = {1, 1, 1};
  aps.intra_lod_prediction_skip_layers = lifting ? 0x7fffffff : 0;
  aps.pred_weight_blending_enabled_flag = false;
  aps.num_detail_levels_minus1 = 9;
  aps.lodSamplingPeriod.assign(10, 4);
  aps.dist2 = 0;
  aps.aps_slice_dist2_deltas_present_flag = false;
  aps.scalable_lifting_enabled_flag = false;
  aps.max_neigh_range_minus1 = 0;
  aps.attrInterPredictionEnabled = false;
  pcc::AttributeBrickHeader abh;
  abh.attr_dist2_delta = 0;
  abh.enableAttrInterPred = false;
  pcc::AttributeInterPredParams inter;
  inter.enableAttrInterPred = false;
  inter.attrInterIntraSliceRDO = false;

  pcc::AttributeLods a, b;
  a.generate(aps, abh, n - 1, 0, cloud, inter);               // the shim
  gpcc_shim::lods_generate_cpu(b, aps, abh, n - 1, 0, cloud, inter);  // reference body

  bool ok = a.numPointsInLod == b.numPointsInLod && a.indexes == b.indexes
    && a.predictors.size() == b.predictors.size();
  long bad = 0;
  for (size_t i = 0; ok && i < a.predictors.size(); i++) {
    const auto &p = a.predictors[i], &q = b.predictors[i];
    bool same = p.neighborCount == q.neighborCount;
    for (uint32_t k = 0; same && k < p.neighborCount; k++)
      same = p.neighbors[k].predictorIndex == q.neighbors[k].predictorIndex
        && p.neighbors[k].weight == q.neighbors[k].weight;
    bad += !same;
  }
  ok = ok && bad == 0;
  gpcc_ctx* ctx = nullptr;
  const bool device = gpcc_ctx_create(0, nullptr, &ctx) == GPCC_OK;
  if (ctx)
    gpcc_ctx_destroy(ctx);
  std::printf(
    "lod_shim_check n=%d lifting=%d lods=%zu path=%s result=%s (%ld predictors differ)\n",
    n, lifting, a.numPointsInLod.size(), device ? "device" : "cpu-fallback",
    ok ? "identical" : "MISMATCH", bad);
  return ok ? 0 : 1;
}
