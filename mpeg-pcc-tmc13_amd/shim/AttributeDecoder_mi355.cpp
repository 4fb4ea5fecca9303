// AttributeDecoder_mi355.cpp -- seam 3, decoder side: the operator behind
//   pcc::makeAttributeDecoder()                       (tmc3/Attribute.h:80,
//                                                      AttributeDecoder.cpp:183-187)
// for the LIFTING and PREDICTING transforms (decodeColorsLift /
// decodeReflectancesLift AttributeDecoder.cpp:678-857, decodeColorsPred /
// decodeReflectancesPred :328-523).  As on the encoder side
// (AttributeEncoder_mi355.cpp) the seam is the factory: the integrator compiles
// tmc3/AttributeDecoder.cpp with -DmakeAttributeDecoder=makeAttributeDecoderCpu
// and adds this translation unit.  The symbols of a slice do not depend on its
// reconstruction, so the whole slice is parsed first -- the reference's own
// arithmetic decoder and context models, the residual syntax of shim_common.hpp --
// and ONE device call (gpcc_lift_decode_attr / gpcc_pred_decode_attr: LoD build +
// inverse transform) turns the values into the attributes.  Since round 4 an intra RAHT
// slice is decoded the same way (decodeColorsRaht / decodeReflectancesRaht
// :613-674, 527-609): symbols parsed, then gpcc_raht_decode_attr_regions -- Morton codes, sort,
// the QP regions' offsets per point (round 5), inverse transform, clip, scatter -- in one device call.
// Every other slice (raw, RAHT with inter prediction -- seam 1 --, a partial geometry octree) goes to the
// reference's decoder unchanged.
//
// Built against the reference's headers; contains no reference code.
#include <memory>
#include <vector>

#include "Attribute.h"

#include "shim_common.hpp"

namespace pcc {
// the reference's factory, renamed at compile time (see above)
std::unique_ptr<AttributeDecoderIntf> makeAttributeDecoderCpu();
}  // namespace pcc

namespace gpcc_shim {
long long g_dec_device = 0, g_dec_cpu = 0;

namespace {

using namespace pcc;

class DeviceAttributeDecoder : public AttributeDecoderIntf {
public:
  DeviceAttributeDecoder() : _cpu(makeAttributeDecoderCpu()) {}

  void decode(
    const SequenceParameterSet& sps, const AttributeDescription& desc,
    const AttributeParameterSet& aps, const AttributeBrickHeader& abh,
    int geom_num_points_minus1, int minGeomNodeSizeLog2, const char* payload,
    size_t payloadLen, AttributeContexts& ctxtMem, PCCPointSet3& cloud,
    AttributeInterPredParams& inter) override
  {
    const bool ours = aps.attr_encoding == AttributeEncoding::kLiftingTransform
      || aps.attr_encoding == AttributeEncoding::kPredictingTransform;
    gpcc_clear_last_error();  // (what a decline of THIS slice reports is this slice's)
    _first.note(aps, abh, inter);  // (shim_common.hpp FirstLods: what the reference's object would cache)
    if (ours) {
      // (scalable lifting: whole slices only -- no points skipped by a partial decode)
      const bool whole =
        !aps.scalable_lifting_enabled_flag || geom_num_points_minus1 + 1 == int(cloud.getPointCount());
      if (whole && on_device(sps, desc, aps, abh, minGeomNodeSizeLog2, payload, payloadLen, ctxtMem, cloud, inter)) {
        g_dec_device++;
        return;
      }
      g_dec_cpu++;
      strict_check("the lifting / predicting attribute decoder");
    }
    if (aps.attr_encoding == AttributeEncoding::kRAHTransform
        && raht_on_device(sps, desc, aps, abh, payload, payloadLen, ctxtMem, cloud, inter)) {
      g_dec_device++;
      return;
    }
    ScopedLodOverride build_as_cached(_first);
    _cpu->decode(
      sps, desc, aps, abh, geom_num_points_minus1, minGeomNodeSizeLog2, payload, payloadLen,
      ctxtMem, cloud, inter);
  }

  bool isReusable(const AttributeParameterSet& aps, const AttributeBrickHeader& abh) const override
  {
    return _first.reusable(aps, abh);
  }

private:
  bool on_device(
    const SequenceParameterSet& sps, const AttributeDescription& desc,
    const AttributeParameterSet& aps, const AttributeBrickHeader& abh,
    int minGeomNodeSizeLog2, const char* payload, size_t payloadLen,
    AttributeContexts& ctxtMem, PCCPointSet3& cloud, AttributeInterPredParams& inter)
  {
    const int c = desc.attr_num_dimensions_minus1 + 1;
    const int n = int(cloud.getPointCount());
    const bool interSlice = inter.enableAttrInterPred;   // (one component: the reflectance drivers)
    if ((c != 1 && c != 3) || n <= 0 || (interSlice && c != 1))
      return false;
    gpcc_ctx* ctx = process_context("the attribute decoder");
    gpcc_lod_params lod;
    if (_first.inter != interSlice)
      return false;
    if (!ctx || !flatten_lod(_first.aps, _first.abh, minGeomNodeSizeLog2, inter, &lod, true))
      return false;
    const QpSet qpSet = deriveQpSet(desc, aps, abh);
    if (interSlice && !qpSet.regions.empty())
      return false;  // (the entries with a reference frame take no QP regions)
    const bool lifting = aps.attr_encoding == AttributeEncoding::kLiftingTransform;
    gpcc_lift_params lp{};
    gpcc_pred_params pp{};
    if (lifting ? !flatten_qp(qpSet, &lp) : !flatten_qp(qpSet, &pp))
      return false;

    // ---- the slice's symbols: the reference's arithmetic decoder, set up as
    //      PCCResidualsDecoder::start (:80-88) does.  A copy of the models is
    //      advanced; the caller's are replaced only when the slice is done ----------
    SliceContexts models(ctxtMem);
    EntropyDecoder ac;
    ac.setBuffer(payloadLen, payload);
    ac.enableBypassStream(sps.cabac_bypass_stream_enabled_flag);
    ac.setBypassBinCodingWithoutProbUpdate(sps.bypass_bin_coding_without_prob_update);
    ac.start();
    std::vector<int32_t> values(size_t(c) * n);
    models.parse_slice(ac, n, c, values.data());
    ac.stop();

    // ---- LoD build + inverse transform -----------------------------------------------
    std::vector<int32_t> xyz, attrs(size_t(c) * n);
    positions_of(cloud, &xyz);
    int rc;
    InterStructure is;
    if (interSlice) {
      rc = build_inter_structure(ctx, lod, xyz, n, _first.abh, inter, &is);
      if (!rc && lifting) {
        lp.bitdepth = desc.bitdepth;
        lp.fixed_point_qp_offset = qpSet.fixedPointQpOffset;
        lp.num_lods = is.nl;
        for (int l = 0; l < is.nl; l++)
          lp.num_points_in_lod[l] = is.npl[l];
        rc = gpcc_lift_inverse_inter(
          ctx, &lp, n, is.nc.data(), is.ni.data(), is.nw.data(), is.xr.data(), is.idx.data(), attrs.data(),
          is.attrsFrame.data(), is.nFrame, values.data());
      } else if (!rc) {
        pp.bitdepth = desc.bitdepth;
        pp.max_num_direct_predictors = aps.max_num_direct_predictors;
        pp.direct_avg_predictor_disabled_flag = aps.direct_avg_predictor_disabled_flag;
        pp.adaptive_prediction_threshold = aps.adaptivePredictionThreshold(desc);
        for (int k = 0; k < 3; k++)
          pp.quant_neigh_weight[k] = aps.quant_neigh_weight[k];
        pp.max_num_detail_levels = aps.maxNumDetailLevels();
        pp.num_lods = is.nl;
        for (int l = 0; l < is.nl; l++)
          pp.num_points_in_lod[l] = is.npl[l];
        rc = gpcc_pred_inverse_inter(
          ctx, &pp, n, is.nc.data(), is.ni.data(), is.nw.data(), is.xr.data(), is.idx.data(), attrs.data(),
          is.attrsFrame.data(), is.nFrame, values.data());
      }
    } else if (lifting) {
      lp.bitdepth = desc.bitdepth;
      lp.fixed_point_qp_offset = qpSet.fixedPointQpOffset;
      lp.last_component_prediction_enabled_flag = abh.lcpPresent(desc, aps);
      int8_t lcp[GPCC_MAX_LODS] = {};
      if (lp.last_component_prediction_enabled_flag)
        for (size_t l = 0; l < abh.attrLcpCoeffs.size() && l < GPCC_MAX_LODS; l++)
          lcp[l] = abh.attrLcpCoeffs[l];
      rc = gpcc_lift_decode_attr(ctx, &lod, &lp, xyz.data(), attrs.data(), values.data(), lcp, nullptr, n, c);
    } else {
      pp.bitdepth = desc.bitdepth;
      pp.max_num_direct_predictors = aps.max_num_direct_predictors;
      pp.direct_avg_predictor_disabled_flag = aps.direct_avg_predictor_disabled_flag;
      pp.adaptive_prediction_threshold = aps.adaptivePredictionThreshold(desc);
      pp.inter_component_prediction_enabled_flag = abh.icpPresent(desc, aps);
      for (int k = 0; k < 3; k++)
        pp.quant_neigh_weight[k] = aps.quant_neigh_weight[k];
      pp.max_num_detail_levels = aps.maxNumDetailLevels();
      int8_t icp[GPCC_MAX_LODS][3] = {};
      if (pp.inter_component_prediction_enabled_flag)
        for (size_t l = 0; l < abh.icpCoeffs.size() && l < GPCC_MAX_LODS; l++)
          for (int k = 0; k < 3; k++)
            icp[l][k] = abh.icpCoeffs[l][k];
      rc = gpcc_pred_decode_attr(ctx, &lod, &pp, xyz.data(), attrs.data(), values.data(), &icp[0][0], nullptr, n, c);
    }
    if (rc) {
      if (rc != GPCC_ERR_UNSUPPORTED)
        std::fprintf(stderr, "gpcc: %s; the attribute decoder falls back to the CPU\n", gpcc_last_error());
      return false;
    }
    store_attributes(attrs, c, &cloud);
    ctxtMem = models.saved();
    return true;
  }

  // ---- an intra RAHT slice -----------------------------------------------------------------
  bool raht_on_device(
    const SequenceParameterSet& sps, const AttributeDescription& desc,
    const AttributeParameterSet& aps, const AttributeBrickHeader& abh, const char* payload,
    size_t payloadLen, AttributeContexts& ctxtMem, PCCPointSet3& cloud, AttributeInterPredParams& inter)
  {
    const int c = desc.attr_num_dimensions_minus1 + 1;
    const int n = int(cloud.getPointCount());
    if ((c != 1 && c != 3) || n <= 0 || inter.enableAttrInterPred)
      return false;
    const QpSet qpSet = deriveQpSet(desc, aps, abh);
    gpcc_raht_params rp;
    gpcc_qp_regions regions;  // (round 5: QP regions, offsets per point derived on the device)
    if (!flatten_regions(qpSet, &regions) || !flatten_raht(aps.rahtPredParams, qpSet, aps.raht_extension, inter, &rp))
      return false;
    gpcc_ctx* ctx = process_context("the attribute decoder");
    if (!ctx)
      return false;
    SliceContexts models(ctxtMem);
    EntropyDecoder ac;
    ac.setBuffer(payloadLen, payload);
    ac.enableBypassStream(sps.cabac_bypass_stream_enabled_flag);
    ac.setBypassBinCodingWithoutProbUpdate(sps.bypass_bin_coding_without_prob_update);
    ac.start();
    // the symbols come point by point in Morton order; the transform reads them planar [c][n]
    std::vector<int32_t> values(size_t(c) * n), coeffs;
    models.parse_slice(ac, n, c, values.data());
    ac.stop();
    const int32_t* co = values.data();
    if (c == 3) {
      coeffs.resize(size_t(c) * n);
      for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
          coeffs[size_t(k) * n + i] = values[size_t(3) * i + k];
      co = coeffs.data();
    }
    std::vector<int32_t> xyz, attrs(size_t(c) * n);
    positions_of(cloud, &xyz);
    const int rc = gpcc_raht_decode_attr_regions(ctx, &rp, &regions, xyz.data(), attrs.data(), co, n, c, desc.bitdepth);
    if (rc) {
      if (rc != GPCC_ERR_UNSUPPORTED)
        std::fprintf(stderr, "gpcc: %s; the attribute decoder falls back to the CPU\n", gpcc_last_error());
      return false;
    }
    store_attributes(attrs, c, &cloud);
    ctxtMem = models.saved();
    return true;
  }

  std::unique_ptr<AttributeDecoderIntf> _cpu;
  FirstLods _first;
};

}  // namespace
}  // namespace gpcc_shim

namespace pcc {
std::unique_ptr<AttributeDecoderIntf>
makeAttributeDecoder()
{
  return std::unique_ptr<AttributeDecoderIntf>(new gpcc_shim::DeviceAttributeDecoder());
}
}  // namespace pcc

extern "C" void
gpcc_shim_decoder_counters(long long out[2])
{
  out[0] = gpcc_shim::g_dec_device;
  out[1] = gpcc_shim::g_dec_cpu;
}
