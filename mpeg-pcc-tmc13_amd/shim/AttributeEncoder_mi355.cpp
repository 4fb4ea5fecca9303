// AttributeEncoder_mi355.cpp -- seam 3, encoder side: the operator behind
//   pcc::makeAttributeEncoder()                       (tmc3/Attribute.h:107,
//                                                      AttributeEncoder.cpp:456-460)
// for the LIFTING and PREDICTING transforms and, since round 4, for RAHT slices.  The
// reference's slice drivers (encodeColorsLift / encodeReflectancesLift
// AttributeEncoder.cpp:1379-1494, 1543-1648; encodeColorsPred / encodeReflectancesPred
// :1075-1210, 749-853; encodeColorsTransformRaht / encodeReflectancesTransformRaht
// :1306-1375, 1214-1302) are private members that interleave transform and entropy coder,
// so they are not a link seam; the FACTORY is.  The integrator compiles tmc3/AttributeEncoder.cpp
// with -DmakeAttributeEncoder=makeAttributeEncoderCpu (the reference class
// stays in the link, reachable through the renamed factory) and adds this
// translation unit, which defines makeAttributeEncoder() with the original
// signature and returns an AttributeEncoderIntf that
//   * for a lifting / predicting slice without QP regions (with attribute inter prediction:
//     one component, no slice-level inter / intra decision -- gpcc_lod_build_inter +
//     gpcc_lift_forward_inter / gpcc_pred_forward_inter)
//     runs LoD build + transform (gpcc_lift_encode_attr / gpcc_pred_encode_attr),
//     zero-run formation (gpcc_zero_run_pack) and binarisation
//     (gpcc_binarise_symbols) on the MI355X, and then replays the binary
//     decisions on the reference's own arithmetic coder and context models --
//     what AttributeEncoder::encode (:466-634) does around its drivers
//     (deriveQpSet, the brick header, the payload, the saved contexts) is
//     repeated here through the reference's public functions;
//   * for an intra RAHT slice without QP regions runs the WHOLE slice driver on the device
//     (gpcc_raht_encode_attr_packed: Morton codes, sort, transform, clip, scatter, zero runs)
//     + gpcc_binarise_symbols, and replays the decisions the same way -- through seam 1 alone
//     the reference's host-side std::sort and entropy loop were 127 of the slice's 143 ms;
//   * hands every other slice (raw, RAHT with inter prediction or QP regions -- which still
//     reach the device's transform through seam 1 --, ...) to the reference's encoder unchanged.
//
// Built against the reference's headers; contains no reference code.
#include <memory>
#include <vector>

#include "Attribute.h"
#include "PayloadBuffer.h"
#include "io_hls.h"

#include "shim_common.hpp"

namespace pcc {
// the reference's factory, renamed at compile time (see above)
std::unique_ptr<AttributeEncoderIntf> makeAttributeEncoderCpu();
}  // namespace pcc

namespace gpcc_shim {
// {slices coded through the device, slices handed to the reference's encoder
//  although they are lifting / predicting slices}
long long g_enc_device = 0, g_enc_cpu = 0;

namespace {

using namespace pcc;

class DeviceAttributeEncoder : public AttributeEncoderIntf {
public:
  DeviceAttributeEncoder() : _cpu(makeAttributeEncoderCpu()) {}

  void encode(
    const SequenceParameterSet& sps, const AttributeDescription& desc,
    const AttributeParameterSet& aps, AttributeBrickHeader& abh,
    AttributeContexts& ctxtMem, PCCPointSet3& cloud, PayloadBuffer* payload,
    AttributeInterPredParams& inter) override
  {
    const bool ours = aps.attr_encoding == AttributeEncoding::kLiftingTransform
      || aps.attr_encoding == AttributeEncoding::kPredictingTransform;
    gpcc_clear_last_error();  // (what a decline of THIS slice reports is this slice's)
    _first.note(aps, abh, inter);  // (the structure this object's reference twin would cache from here on)
    if (ours) {
      if (on_device(sps, desc, aps, abh, ctxtMem, cloud, payload, inter)) {
        g_enc_device++;
        return;
      }
      g_enc_cpu++;
      strict_check("the lifting / predicting attribute encoder");
    }
    // RAHT: declined slices (inter prediction) are not counted as fall-backs of
    // this seam -- the reference's driver then calls seam 1, which keeps its own counters
    if (aps.attr_encoding == AttributeEncoding::kRAHTransform
        && raht_on_device(sps, desc, aps, abh, ctxtMem, cloud, payload, inter)) {
      g_enc_device++;
      return;
    }
    ScopedLodOverride build_as_cached(_first);
    _cpu->encode(sps, desc, aps, abh, ctxtMem, cloud, payload, inter);
  }

  bool isReusable(const AttributeParameterSet& aps, const AttributeBrickHeader& abh) const override
  {
    return _first.reusable(aps, abh);
  }

private:
  bool on_device(
    const SequenceParameterSet& sps, const AttributeDescription& desc,
    const AttributeParameterSet& aps, AttributeBrickHeader& abh,
    AttributeContexts& ctxtMem, PCCPointSet3& cloud, PayloadBuffer* payload,
    AttributeInterPredParams& inter)
  {
    const int c = desc.attr_num_dimensions_minus1 + 1;
    const int n = int(cloud.getPointCount());
    // attribute inter prediction: one component, and without the slice-level inter / intra
    // decision (which codes the slice twice and compares, AttributeEncoder.cpp:520-585)
    const bool interSlice = inter.enableAttrInterPred;
    if ((c != 1 && c != 3) || n <= 0 || inter.codeAttributeSecondPass() || (interSlice && c != 1))
      return false;
    gpcc_ctx* ctx = process_context("the attribute encoder");
    gpcc_lod_params lod;
    // the structure: of the parameters the reference's cache was (would have been) built with
    if (_first.inter != interSlice)
      return false;
    if (!ctx || !flatten_lod(_first.aps, _first.abh, 0, inter, &lod, true))
      return false;
    const QpSet qpSet = deriveQpSet(desc, aps, abh);
    if (interSlice && !qpSet.regions.empty())
      return false;  // (the entries with a reference frame take no QP regions)
    const bool lifting = aps.attr_encoding == AttributeEncoding::kLiftingTransform;

    std::vector<int32_t> xyz, attrs, values(size_t(c) * n);
    positions_of(cloud, &xyz);
    attributes_of(cloud, c, &attrs);

    // ---- LoD build + transform: the values of every predictor in coding order ----
    int8_t lcp[GPCC_MAX_LODS] = {};
    int8_t icp[GPCC_MAX_LODS][3] = {};
    InterStructure is;
    if (interSlice && build_inter_structure(ctx, lod, xyz, n, _first.abh, inter, &is))
      return declined();
    if (interSlice && lifting) {
      gpcc_lift_params lp{};
      if (!flatten_qp(qpSet, &lp))
        return false;
      lp.bitdepth = desc.bitdepth;
      lp.fixed_point_qp_offset = qpSet.fixedPointQpOffset;
      lp.num_lods = is.nl;
      for (int l = 0; l < is.nl; l++)
        lp.num_points_in_lod[l] = is.npl[l];
      if (gpcc_lift_forward_inter(
            ctx, &lp, n, is.nc.data(), is.ni.data(), is.nw.data(), is.xr.data(), is.idx.data(), attrs.data(),
            is.attrsFrame.data(), is.nFrame, values.data()))
        return declined();
    } else if (interSlice) {
      gpcc_pred_params pp{};
      if (!flatten_qp(qpSet, &pp))
        return false;
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
      if (gpcc_pred_forward_inter(
            ctx, &pp, n, is.nc.data(), is.ni.data(), is.nw.data(), is.xr.data(), is.idx.data(), attrs.data(),
            is.attrsFrame.data(), is.nFrame, values.data()))
        return declined();
    } else if (lifting) {
      gpcc_lift_params lp{};
      if (!flatten_qp(qpSet, &lp))
        return false;
      lp.bitdepth = desc.bitdepth;
      lp.fixed_point_qp_offset = qpSet.fixedPointQpOffset;
      lp.last_component_prediction_enabled_flag = abh.lcpPresent(desc, aps);
      if (gpcc_lift_encode_attr(ctx, &lod, &lp, xyz.data(), attrs.data(), values.data(), lcp, nullptr, n, c))
        return declined();
    } else {
      gpcc_pred_params pp{};
      if (!flatten_qp(qpSet, &pp))
        return false;
      pp.bitdepth = desc.bitdepth;
      pp.max_num_direct_predictors = aps.max_num_direct_predictors;
      pp.direct_avg_predictor_disabled_flag = aps.direct_avg_predictor_disabled_flag;
      pp.adaptive_prediction_threshold = aps.adaptivePredictionThreshold(desc);
      pp.inter_component_prediction_enabled_flag = abh.icpPresent(desc, aps);
      for (int k = 0; k < 3; k++)
        pp.quant_neigh_weight[k] = aps.quant_neigh_weight[k];
      pp.max_num_detail_levels = aps.maxNumDetailLevels();
      if (gpcc_pred_encode_attr(ctx, &lod, &pp, xyz.data(), attrs.data(), values.data(), &icp[0][0], nullptr, n, c))
        return declined();
    }

    // ---- zero runs and binary decisions, on the device where the values are ------
    std::vector<int32_t> runs(n), syms(size_t(c) * n);
    int32_t num_symbols = 0, trailing = 0;
    if (gpcc_zero_run_pack(ctx, values.data(), n, c, 0, runs.data(), syms.data(), &num_symbols, &trailing))
      return declined();
    int64_t num_bins = 0;
    std::vector<uint8_t> bins(size_t(n) * (c == 3 ? 24 : 12) + 1024);
    int rc = gpcc_binarise_symbols(
      ctx, runs.data(), syms.data(), num_symbols, trailing, c, bins.data(), int64_t(bins.size()), &num_bins);
    if (rc && num_bins > int64_t(bins.size())) {
      bins.resize(size_t(num_bins));
      rc = gpcc_binarise_symbols(
        ctx, runs.data(), syms.data(), num_symbols, trailing, c, bins.data(), int64_t(bins.size()), &num_bins);
    }
    if (rc)
      return declined();

    // ---- from here on nothing can decline: the slice header may be written ---------
    if (lifting && abh.lcpPresent(desc, aps))
      abh.attrLcpCoeffs.assign(lcp, lcp + aps.maxNumDetailLevels());
    if (!lifting && abh.icpPresent(desc, aps)) {
      abh.icpCoeffs.resize(aps.maxNumDetailLevels());
      for (int l = 0; l < aps.maxNumDetailLevels(); l++)
        abh.icpCoeffs[l] = Vec3<int8_t>{icp[l][0], icp[l][1], icp[l][2]};
    }

    // the reference's arithmetic coder, set up as PCCResidualsEncoder::start
    // (:113-121) does, driven by the decisions
    SliceContexts models(ctxtMem);
    EntropyEncoder ac;
    ac.setBuffer(n * 3 * 2 + 1024, nullptr);
    ac.enableBypassStream(sps.cabac_bypass_stream_enabled_flag);
    ac.setBypassBinCodingWithoutProbUpdate(sps.bypass_bin_coding_without_prob_update);
    ac.start();
    for (int64_t i = 0; i < num_bins; i++) {
      const int id = bins[i] >> 1, bin = bins[i] & 1;
      if (id == 31)
        ac.encode(bin);
      else
        ac.encode(bin, models.model(id));
    }
    const uint32_t len = ac.stop();

    abh.RAHTFilterTaps.assign(
      inter.paramsForInterRAHT.FilterTaps.begin(), inter.paramsForInterRAHT.FilterTaps.end());
    write(sps, aps, abh, payload);
    payload->insert(payload->end(), ac.buffer(), ac.buffer() + len);
    ctxtMem = models.saved();
    store_attributes(attrs, c, &cloud);
    // (the reference's reflectance drivers leave the slice's distortion estimate for the slice-level
    // inter / intra decision here, AttributeEncoder.cpp:760, 826, 1554; such slices are declined above --
    // codeAttributeSecondPass -- so nothing reads it: zero, not a value of an earlier slice)
    if (c == 1)
      inter.distEstimate = 0.;
    return true;
  }

  // ---- an intra RAHT slice: what AttributeEncoder::encode (:466-634) does around
  //      encode{Colors,Reflectances}TransformRaht, with the driver itself on the device ---------
  bool raht_on_device(
    const SequenceParameterSet& sps, const AttributeDescription& desc,
    const AttributeParameterSet& aps, AttributeBrickHeader& abh,
    AttributeContexts& ctxtMem, PCCPointSet3& cloud, PayloadBuffer* payload,
    AttributeInterPredParams& inter)
  {
    const int c = desc.attr_num_dimensions_minus1 + 1;
    const int n = int(cloud.getPointCount());
    if ((c != 1 && c != 3) || n <= 0 || inter.enableAttrInterPred || inter.codeAttributeSecondPass())
      return false;
    const QpSet qpSet = deriveQpSet(desc, aps, abh);
    gpcc_raht_params rp;
    // (QP regions, round 5: the device derives every point's offset from its position, as
    // qpSet.regionQpOffset does in the reference's drivers, AttributeEncoder.cpp:1262, 1336)
    gpcc_qp_regions regions;
    if (!flatten_regions(qpSet, &regions) || !flatten_raht(aps.rahtPredParams, qpSet, aps.raht_extension, inter, &rp))
      return false;
    gpcc_ctx* ctx = process_context("the attribute encoder");
    if (!ctx)
      return false;
    std::vector<int32_t> xyz, attrs;
    positions_of(cloud, &xyz);
    attributes_of(cloud, c, &attrs);
    std::vector<int32_t> runs(n), syms(size_t(c) * n);
    int32_t num_symbols = 0, trailing = 0;
    if (gpcc_raht_encode_attr_packed_regions(
          ctx, &rp, &regions, xyz.data(), attrs.data(), runs.data(), syms.data(), &num_symbols, &trailing, n, c,
          desc.bitdepth))
      return declined();
    int64_t num_bins = 0;
    std::vector<uint8_t> bins(size_t(num_symbols) * (c == 3 ? 24 : 12) + 1024);
    int rc = gpcc_binarise_symbols(
      ctx, runs.data(), syms.data(), num_symbols, trailing, c, bins.data(), int64_t(bins.size()), &num_bins);
    if (rc && num_bins > int64_t(bins.size())) {
      bins.resize(size_t(num_bins));
      rc = gpcc_binarise_symbols(
        ctx, runs.data(), syms.data(), num_symbols, trailing, c, bins.data(), int64_t(bins.size()), &num_bins);
    }
    if (rc)
      return declined();

    // ---- nothing can decline any more: header, payload, contexts, reconstruction ---------
    if (c == 1)
      inter.paramsForInterRAHT.FilterTaps.clear();  // (:1236: the reflectance driver does, the colour one does not)
    abh.raht_attr_layer_code_mode = inter.attr_layer_code_mode;
    SliceContexts models(ctxtMem);
    EntropyEncoder ac;
    ac.setBuffer(n * 3 * 2 + 1024, nullptr);
    ac.enableBypassStream(sps.cabac_bypass_stream_enabled_flag);
    ac.setBypassBinCodingWithoutProbUpdate(sps.bypass_bin_coding_without_prob_update);
    ac.start();
    for (int64_t i = 0; i < num_bins; i++) {
      const int id = bins[i] >> 1, bin = bins[i] & 1;
      if (id == 31)
        ac.encode(bin);
      else
        ac.encode(bin, models.model(id));
    }
    const uint32_t len = ac.stop();
    abh.RAHTFilterTaps.assign(
      inter.paramsForInterRAHT.FilterTaps.begin(), inter.paramsForInterRAHT.FilterTaps.end());
    write(sps, aps, abh, payload);
    payload->insert(payload->end(), ac.buffer(), ac.buffer() + len);
    ctxtMem = models.saved();
    store_attributes(attrs, c, &cloud);
    return true;
  }

  static bool declined()
  {
    if (gpcc_last_error()[0])
      std::fprintf(stderr, "gpcc: %s; the attribute encoder falls back to the CPU\n", gpcc_last_error());
    return false;
  }

  std::unique_ptr<AttributeEncoderIntf> _cpu;
  FirstLods _first;
};

}  // namespace
}  // namespace gpcc_shim

namespace pcc {
std::unique_ptr<AttributeEncoderIntf>
makeAttributeEncoder()
{
  return std::unique_ptr<AttributeEncoderIntf>(new gpcc_shim::DeviceAttributeEncoder());
}
}  // namespace pcc

extern "C" void
gpcc_shim_encoder_counters(long long out[2])
{
  out[0] = gpcc_shim::g_enc_device;
  out[1] = gpcc_shim::g_enc_cpu;
}
