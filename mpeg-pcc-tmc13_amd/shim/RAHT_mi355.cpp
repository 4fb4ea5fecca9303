// RAHT_mi355.cpp -- drop-in replacement translation unit for the reference's
// tmc3/RAHT.cpp link seam.
//
// It defines the two free functions every caller of the reference binds to
//   pcc::regionAdaptiveHierarchicalTransform         (tmc3/RAHT.h:47-57)
//   pcc::regionAdaptiveHierarchicalInverseTransform  (tmc3/RAHT.h:59-69)
// (call sites: AttributeEncoder.cpp:1273,1341; AttributeDecoder.cpp:595,658)
// with the reference's own signatures, flattens the reference's parameter
// structs into the POD block of include/gpcc_attr_mi355.h and runs the slice
// on the MI355X through the C ABI -- slices with attribute inter prediction
// (attrInterPredParams.enableAttrInterPred) through gpcc_raht_forward_inter /
// gpcc_raht_inverse_inter with the reference frame of paramsForInterRAHT, the
// per-layer modes and filter taps appended to / read from the same vectors the
// reference's function uses (RAHT.cpp:1293, 1820-1825; :1260, 1303).  When the
// device path declines a slice (GPCC_ERR_UNSUPPORTED: inter prediction under the
// integer Haar kernel when the two trees do not line up on octree levels)
// or no GPU is present it calls the reference's CPU implementation, which
// the integrator keeps in the link under a suffixed name (see
// RAHT.cpp is compiled with
//   -DregionAdaptiveHierarchicalTransform=regionAdaptiveHierarchicalTransformCpu
//   -DregionAdaptiveHierarchicalInverseTransform=regionAdaptiveHierarchicalInverseTransformCpu).
//
// Built against the reference's headers; contains no reference code.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "FixedPoint.h"
#include "PCCTMC3Common.h"
#include "hls.h"
#include "quantization.h"

#include "gpcc_attr_mi355.h"

namespace pcc {

// the reference implementation, renamed at compile time (see above)
void regionAdaptiveHierarchicalTransformCpu(
  const RahtPredictionParams& rahtPredParams, const QpSet& qpset,
  const Qps* pointQPOffset, int64_t* mortonCode, int* attributes,
  const int attribCount, const int voxelCount, int* coefficients,
  const bool removeRoundingOps, AttributeInterPredParams& attrInterPredParam);

void regionAdaptiveHierarchicalInverseTransformCpu(
  const RahtPredictionParams& rahtPredParams, const QpSet& qpset,
  const Qps* pointQpOffset, int64_t* mortonCode, int* attributes,
  const int attribCount, const int voxelCount, int* coefficients,
  const bool removeRoundingOps, AttributeInterPredParams& attrInterPredParams);

namespace {

// What this TU did, for an integrator's log and for the drop-in tests: a
// silent fallback must not be mistaken for the device path.
long long g_device_calls = 0, g_cpu_calls = 0;

// GPCC_STRICT=1: a slice that cannot run on the device is an error, not a
// CPU fallback (CI on a GPU box)
void
cpu_fallback(const char* what)
{
  g_cpu_calls++;
  const char* strict = std::getenv("GPCC_STRICT");
  if (strict && strict[0] == '1') {
    std::fprintf(stderr, "gpcc: GPCC_STRICT=1 and %s did not run on the device (%s)\n", what, gpcc_last_error());
    std::abort();
  }
}

gpcc_ctx*
device_context()
{
  // one context per process; the reference is single threaded
  static gpcc_ctx* ctx = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // (the parameter blocks' sizes belong to the ABI version: see shim_common.hpp process_context)
    if (gpcc_abi_version() != GPCC_ABI_VERSION) {
      std::fprintf(
        stderr, "gpcc: libgpcc_attr_mi355 has ABI %d, this binary was built against %d; RAHT stays on the CPU\n",
        gpcc_abi_version(), GPCC_ABI_VERSION);
      return nullptr;
    }
    const char* dev = std::getenv("GPCC_DEVICE");
    if (gpcc_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &ctx) != GPCC_OK) {
      std::fprintf(
        stderr, "gpcc: no MI355X context (%s); RAHT stays on the CPU\n",
        gpcc_last_error());
      ctx = nullptr;
    }
    // GPCC_RESERVE_POINTS=n: the sequence's largest slice, known to whoever starts the codec -- everything the
    // transforms would allocate on demand is reserved once, before the first slice (gpcc_ctx_reserve)
    const char* rsv = std::getenv("GPCC_RESERVE_POINTS");
    if (ctx && rsv && std::atoll(rsv) > 0 && gpcc_ctx_reserve(ctx, std::atoll(rsv), 1, 3) != GPCC_OK)
      std::fprintf(stderr, "gpcc: GPCC_RESERVE_POINTS=%s not reserved (%s); workspace grows on demand\n", rsv, gpcc_last_error());
  }
  return ctx;
}

// false: the block cannot express these parameters -> CPU path
bool
flatten(
  const RahtPredictionParams& rp, const QpSet& qs, bool extension,
  const AttributeInterPredParams& inter, gpcc_raht_params* p)
{
  if (rp.predWeightParent.size() != 19)
    return false;
  if (
    rp.raht_subnode_prediction_enabled_flag && rp.predWeightChild.size() != 12)
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
  p->raht_subnode_prediction_enabled_flag =
    rp.raht_subnode_prediction_enabled_flag;
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

// region offsets are all zero in every CTC configuration: skip the upload
const int32_t*
qp_offsets_or_null(const Qps* q, int n)
{
  static_assert(sizeof(Qps) == 2 * sizeof(int32_t), "Qps is two ints");
  for (int i = 0; i < n; i++)
    if (q[i][0] | q[i][1])
      return reinterpret_cast<const int32_t*>(q);
  return nullptr;
}

// the tools of attribute inter prediction (AttributeInterPredParamsForRAHT, PCCTMC3Common.h:236-250)
gpcc_raht_inter_params
inter_tools(const AttributeInterPredParams& inter)
{
  const auto& ir = inter.paramsForInterRAHT;
  gpcc_raht_inter_params ip;
  ip.raht_inter_prediction_depth_minus1 = ir.raht_inter_prediction_depth_minus1;
  ip.raht_enable_inter_intra_layer_rdo = ir.raht_enable_inter_intra_layer_RDO;
  ip.enable_filter_estimation = ir.enableFilterEstimation;
  ip.skip_init_layers_for_filtering = ir.skipInitLayersForFiltering;
  return ip;
}

static_assert(sizeof(int) == sizeof(int32_t), "the reference's int vectors are handed over as int32_t");

}  // namespace

void
regionAdaptiveHierarchicalTransform(
  const RahtPredictionParams& rahtPredParams, const QpSet& qpset,
  const Qps* pointQpOffsets, int64_t* mortonCode, int* attributes,
  const int attribCount, const int voxelCount, int* coefficients,
  const bool rahtExtension, AttributeInterPredParams& attrInterPredParams)
{
  gpcc_raht_params p;
  gpcc_ctx* ctx = device_context();
  if (
    ctx && voxelCount > 0
    && flatten(rahtPredParams, qpset, rahtExtension, attrInterPredParams, &p)) {
    int rc;
    if (attrInterPredParams.enableAttrInterPred) {
      auto& ir = attrInterPredParams.paramsForInterRAHT;
      const gpcc_raht_inter_params ip = inter_tools(attrInterPredParams);
      int32_t modes[32], taps[32], num_modes = 0, num_taps = 0;
      rc = ir.voxelCount <= 0
        ? GPCC_ERR_UNSUPPORTED
        : gpcc_raht_forward_inter(
            ctx, &p, &ip, mortonCode, qp_offsets_or_null(pointQpOffsets, voxelCount), attributes, coefficients, voxelCount,
            attribCount, ir.mortonCode.data(),
            reinterpret_cast<const int32_t*>(ir.attributes.data()), ir.voxelCount, modes, &num_modes, taps, &num_taps);
      if (rc == GPCC_OK) {
        for (int i = 0; i < num_modes; i++)
          attrInterPredParams.attr_layer_code_mode.push_back(modes[i]);
        for (int i = 0; i < num_taps; i++)
          ir.FilterTaps.push_back(taps[i]);
      }
    } else {
      rc = gpcc_raht_forward(
        ctx, &p, mortonCode, qp_offsets_or_null(pointQpOffsets, voxelCount),
        attributes, coefficients, voxelCount, attribCount);
    }
    if (rc == GPCC_OK) {
      g_device_calls++;
      return;
    }
    if (rc != GPCC_ERR_UNSUPPORTED)
      std::fprintf(stderr, "gpcc: %s; slice falls back to the CPU\n", gpcc_last_error());
  }
  cpu_fallback("regionAdaptiveHierarchicalTransform");
  regionAdaptiveHierarchicalTransformCpu(
    rahtPredParams, qpset, pointQpOffsets, mortonCode, attributes,
    attribCount, voxelCount, coefficients, rahtExtension, attrInterPredParams);
}

void
regionAdaptiveHierarchicalInverseTransform(
  const RahtPredictionParams& rahtPredParams, const QpSet& qpset,
  const Qps* pointQpOffsets, int64_t* mortonCode, int* attributes,
  const int attribCount, const int voxelCount, int* coefficients,
  const bool rahtExtension, AttributeInterPredParams& attrInterPredParams)
{
  gpcc_raht_params p;
  gpcc_ctx* ctx = device_context();
  if (
    ctx && voxelCount > 0
    && flatten(rahtPredParams, qpset, rahtExtension, attrInterPredParams, &p)) {
    int rc;
    if (attrInterPredParams.enableAttrInterPred) {
      const auto& ir = attrInterPredParams.paramsForInterRAHT;
      const gpcc_raht_inter_params ip = inter_tools(attrInterPredParams);
      const auto& modes = attrInterPredParams.attr_layer_code_mode;
      rc = ir.voxelCount <= 0 || modes.size() > 32 || ir.FilterTaps.size() > 32
        ? GPCC_ERR_UNSUPPORTED
        : gpcc_raht_inverse_inter(
            ctx, &p, &ip, mortonCode, qp_offsets_or_null(pointQpOffsets, voxelCount), attributes, coefficients, voxelCount,
            attribCount, ir.mortonCode.data(),
            reinterpret_cast<const int32_t*>(ir.attributes.data()), ir.voxelCount,
            reinterpret_cast<const int32_t*>(modes.data()), int32_t(modes.size()),
            reinterpret_cast<const int32_t*>(ir.FilterTaps.data()), int32_t(ir.FilterTaps.size()));
    } else {
      rc = gpcc_raht_inverse(
        ctx, &p, mortonCode, qp_offsets_or_null(pointQpOffsets, voxelCount),
        attributes, coefficients, voxelCount, attribCount);
    }
    if (rc == GPCC_OK) {
      g_device_calls++;
      return;
    }
    if (rc != GPCC_ERR_UNSUPPORTED)
      std::fprintf(stderr, "gpcc: %s; slice falls back to the CPU\n", gpcc_last_error());
  }
  cpu_fallback("regionAdaptiveHierarchicalInverseTransform");
  regionAdaptiveHierarchicalInverseTransformCpu(
    rahtPredParams, qpset, pointQpOffsets, mortonCode, attributes,
    attribCount, voxelCount, coefficients, rahtExtension, attrInterPredParams);
}

}  // namespace pcc

// {calls that ran on the device, calls handed to the reference's CPU function}
extern "C" void
gpcc_shim_raht_counters(long long out[2])
{
  out[0] = pcc::g_device_calls;
  out[1] = pcc::g_cpu_calls;
}
