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
// device path declines (no GPU, a partially decoded scalable slice, ...) the renamed reference body runs instead.
// With attribute inter prediction the build is gpcc_lod_build_inter.
//
// Built against the reference's headers; contains no reference code.
#include <vector>

#include "shim_common.hpp"

namespace gpcc_shim {
// AttributeLods_cpu_adapter.cpp: calls the renamed reference member
void lods_generate_cpu(
  pcc::AttributeLods& lods, const pcc::AttributeParameterSet& aps,
  const pcc::AttributeBrickHeader& abh, int geom_num_points_minus1,
  int minGeomNodeSizeLog2, const pcc::PCCPointSet3& cloud,
  const pcc::AttributeInterPredParams& attrInterPredParams);

// what this TU did (gpcc_shim_lod_counters)
long long g_lod_device_calls = 0, g_lod_cpu_calls = 0;
}  // namespace gpcc_shim

namespace pcc {

void
AttributeLods::generate(
  const AttributeParameterSet& aps_in, const AttributeBrickHeader& abh_in,
  int geom_num_points_minus1, int minGeomNodeSizeLog2,
  const PCCPointSet3& cloud, const AttributeInterPredParams& attrInterPredParams)
{
  // (seam 3 hands a slice to the reference's coder whose cache is empty although its object has
  // coded a LoD-based attribute on the device: the structure of THAT attribute's parameters,
  // shim_common.hpp FirstLods)
  const auto& ov = gpcc_shim::lod_override();
  const AttributeParameterSet& aps = ov.aps ? *ov.aps : aps_in;
  const AttributeBrickHeader& abh = ov.abh ? *ov.abh : abh_in;
  gpcc_lod_params lp;
  gpcc_ctx* ctx = gpcc_shim::process_context("the LoD build");
  const int n = int(cloud.getPointCount());
  // (scalable lifting: whole slices only -- no points skipped by a partial decode)
  const bool whole = !aps.scalable_lifting_enabled_flag || geom_num_points_minus1 + 1 == n;
  // attribute inter prediction: the search also looks into the reference frame
  // (gpcc_lod_build_inter); the transforms over the structure stay the reference's
  const bool inter = attrInterPredParams.enableAttrInterPred;
  const int nFrame = inter ? int(attrInterPredParams.referencePointCloud.getPointCount()) : 0;
  if (
    ctx && n > 0 && whole && (!inter || nFrame > 0)
    && gpcc_shim::flatten_lod(aps, abh, minGeomNodeSizeLog2, attrInterPredParams, &lp, true)) {
    std::vector<int32_t> xyz;
    gpcc_shim::positions_of(cloud, &xyz);
    std::vector<int32_t> nc(n), ni(size_t(n) * 3), nw(size_t(n) * 3), idx(n), xr;
    int32_t npl[GPCC_MAX_LODS], nl = 0;
    int rc;
    if (inter) {
      std::vector<int32_t> xyzFrame;
      gpcc_shim::positions_of(attrInterPredParams.referencePointCloud, &xyzFrame);
      xr.resize(size_t(n) * 3);
      rc = gpcc_lod_build_inter(
        ctx, &lp, xyz.data(), n, xyzFrame.data(), nFrame, abh.attrInterPredSearchRange,
        attrInterPredParams.frameDistance, nc.data(), ni.data(), nw.data(), idx.data(), npl, &nl,
        xr.data());
    } else
      rc = gpcc_lod_build(
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
          const bool inFrame = inter && xr[3 * size_t(i) + k] != 0;
          p.neighbors[k].predictorIndex = ni[3 * size_t(i) + k];
          p.neighbors[k].weight = nw[3 * size_t(i) + k];
          // (updatePredictors PCCTMC3Common.h:2286-2293: pointIndex is the neighbour's point --
          // of the reference frame, which predictorIndex then names too, for such a neighbour)
          p.neighbors[k].pointIndex =
            k < nc[i] ? (inFrame ? ni[3 * size_t(i) + k] : idx[ni[3 * size_t(i) + k]]) : 0;
          p.neighbors[k].interFrameRef = inFrame;
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
  gpcc_shim::strict_check("AttributeLods::generate");
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
