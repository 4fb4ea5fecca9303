/* gpcc_attr_mi355.h -- C ABI of the MI355X attribute-transform library.
 *
 * Drop-in boundary for the attribute-transform hot path of TMC13
 * (reference: MPEGGroup/mpeg-pcc-tmc13 @ release-23.0-rc2).  Every entry
 * point names the reference interface it replaces (file:line relative to
 * the reference tree).  Plain pointers and sizes only; no STL, no torch
 * types.  All entry points return 0 on success and a negative
 * gpcc_status on failure; on failure no output buffer holds a valid
 * result and the caller is expected to run the reference CPU function.
 *
 * Two tiers:
 *   - host tier   (gpcc_raht_forward / gpcc_raht_inverse / gpcc_attr_*):
 *     synchronous, caller-owned HOST buffers, exactly the call shape of the
 *     reference free functions, so a replacement translation unit can
 *     forward to it.
 *   - device tier (gpcc_dev_*): the same operations on buffers already
 *     resident in HBM, batched over slices, asynchronous on a HIP stream.
 *     The host tier is implemented on top of it.
 *
 * Limits: at most GPCC_MAX_POINTS points per call (per batch for the device
 * tier) and coordinates in [0, 2^21); beyond them GPCC_ERR_INVALID_ARG.
 */
#ifndef GPCC_ATTR_MI355_H
#define GPCC_ATTR_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 5): gpcc_lift_params / gpcc_pred_params grew by the qp_region_* fields (260 bytes each), GPCC_ERR_RANGE and the
 * inter-frame RAHT entries were added.  A caller compares gpcc_abi_version() with this constant before its first call
 * (the shim TUs do, shim/shim_common.hpp process_context): a library built from another header would read parameter
 * blocks of another size. */
#define GPCC_ABI_VERSION 6
#define GPCC_MAX_POINTS (1 << 29) /* 32-bit device indices, stride <= 3 */

#define GPCC_MAX_QP_LAYERS 32
#define GPCC_MAX_QP_REGIONS 8   /* attr_num_regions a slice header can carry here */
#define GPCC_MAX_AC_QP_LAYERS 32

typedef enum gpcc_status {
  GPCC_OK = 0,
  GPCC_ERR_INVALID_ARG = -1,   /* null pointer, n < 0, c not in {1,2,3}, ... */
  GPCC_ERR_UNSUPPORTED = -2,   /* parameter combination kept on the CPU path */
  GPCC_ERR_NO_DEVICE = -3,     /* no gfx950 device / HIP runtime failure     */
  GPCC_ERR_OUT_OF_MEMORY = -4,
  GPCC_ERR_HIP = -5,           /* a HIP call failed; see gpcc_last_error()   */
  GPCC_ERR_UNSORTED = -6,      /* Morton codes not ascending                 */
  GPCC_ERR_RANGE = -7          /* device tier: values left the range of the fast arithmetic path
                                  (gpcc_ctx_set_fast_arith); nothing was written */
} gpcc_status;

/* Flattened RahtPredictionParams (hls.h:439-466) + QpSet
 * (quantization.h:124-139) + the raht_extension flag
 * (AttributeParameterSet, hls.h:782-876), exactly the values
 * AttributeEncoder.cpp:1273/1341 and AttributeDecoder.cpp:595/658 hand to
 * regionAdaptiveHierarchical{,Inverse}Transform. */
typedef struct gpcc_raht_params {
  int32_t raht_prediction_enabled_flag;
  int32_t integer_haar_enable_flag;
  int32_t raht_prediction_threshold0;
  int32_t raht_prediction_threshold1;
  int32_t raht_subnode_prediction_enabled_flag;
  int32_t raht_prediction_search_range;
  int32_t pred_weight_parent[19]; /* RahtPredictionParams::predWeightParent */
  int32_t pred_weight_child[12];  /* RahtPredictionParams::predWeightChild  */
  int32_t raht_extension;         /* aps.raht_extension                     */

  int32_t num_qp_layers;                      /* QpSet::layers.size() >= 1  */
  int32_t layer_qp[GPCC_MAX_QP_LAYERS][2];    /* QpSet::layers[i] = {luma,
                                                 chroma offset}             */
  int32_t max_qp;                             /* QpSet::maxQp               */
  int32_t fixed_point_qp_offset;              /* QpSet::fixedPointQpOffset  */
  int32_t num_ac_qp_layers;                   /* QpSet::rahtAcCoeffQps.size */
  int32_t ac_qp_offset[GPCC_MAX_AC_QP_LAYERS][7][2];
} gpcc_raht_params;

/* Fill pred_weight_parent / pred_weight_child from the five signalled
 * raht_prediction_weights, as RahtPredictionParams::setPredictionWeights
 * does (hls.h:456-465). */
void gpcc_raht_set_prediction_weights(gpcc_raht_params* p, const int32_t w[5]);

/* ------------------------------------------------------------------ */
/* library / device management                                         */

int gpcc_abi_version(void);
/* Human readable description of the most recent failure on this thread. */
const char* gpcc_last_error(void);
/* Forgets the calling thread's message (a caller that reports "the last error" of a sequence of calls
 * clears it in front of the sequence: the shim TUs do, so that a decline without a message of its own
 * never repeats an earlier slice's). */
void gpcc_clear_last_error(void);
/* Number of usable gfx950 devices (0 if none / no HIP runtime). */
int gpcc_device_count(void);

typedef struct gpcc_ctx gpcc_ctx; /* opaque: device, stream, workspace */

/* Create a context bound to HIP device `device`.  `stream` is a
 * hipStream_t passed as void*: NULL = the library creates its own
 * non-blocking stream; to run on the legacy default stream (the one a null
 * hipStream_t means in a launch) pass GPCC_STREAM_LEGACY, HIP's
 * hipStreamLegacy handle.  Work the caller queues on ANOTHER stream is not
 * ordered against the context's.  Workspace grows on demand and is reused. */
#define GPCC_STREAM_LEGACY ((void*)1)
int gpcc_ctx_create(int device, void* stream, gpcc_ctx** out);
void gpcc_ctx_destroy(gpcc_ctx* ctx);
/* Block until all work queued by this context has completed. */
int gpcc_ctx_synchronize(gpcc_ctx* ctx);
/* Bytes of HBM currently held by the context's workspace. */
size_t gpcc_ctx_workspace_bytes(const gpcc_ctx* ctx);
/* Reserve, ahead of the first transform, everything the RAHT entries allocate on demand for
 * calls of up to `max_points` points in up to `max_slices` slices with `max_c` attribute
 * components at the context's current Morton-bits hint (set that first): the device workspace
 * of the largest flag combination, the pooled blocks and the pinned staging buffers; the
 * device memory is written once and the call returns with the device idle.  A codec that
 * creates a context per sequence and then calls once per (slice, attribute) -- the
 * reference's call pattern, tmc3/AttributeEncoder.cpp:1273, 1341 -- calls this once with its
 * largest slice: no transform allocates afterwards (gpcc_ctx_workspace_bytes stays put), so the
 * first call is not followed by the tens of milliseconds a fresh allocation can cost the calls
 * right behind it.  Not needed for correctness: workspace still grows on
 * demand when a call is larger than what was reserved. */
int gpcc_ctx_reserve(gpcc_ctx* ctx, int64_t max_points, int32_t max_slices, int32_t max_c);
/* Device-tier hint: number of significant Morton-code bits (3 x coordinate
 * bits) of the batches that follow; 0 = unknown (63).  Bounds the number of
 * octree levels that are launched; the host tier derives it itself. */
int gpcc_ctx_set_morton_bits(gpcc_ctx* ctx, int32_t bits);

/* The sub-node prediction kernels (the reference's default flags) compute in doubles where doubles
 * are exact -- every product of the Q15 transform below 2^53: attributes of the bit depth max_qp
 * states (51 + 6 (B - 8), tmc3/quantization.cpp:151) in slices with 2 B + ceil(log2 n) <= 36 -- and
 * in int64 otherwise; the results are the same bits (csrc/raht_arith.hpp).  The kernels check the
 * magnitudes: a call whose values leave the range (attributes wider than max_qp says, a decoder fed
 * arbitrary coefficients) writes nothing; the host tier then repeats it in int64 by itself, the
 * device tier reports GPCC_ERR_RANGE at the next synchronisation.  on = 0: int64 always
 * (also GPCC_F64=0 in the environment).  Default: on. */
int gpcc_ctx_set_fast_arith(gpcc_ctx* ctx, int32_t on);

/* What the context's entries have done since it was created -- lets an
 * integrator (and tests/test_shim_dropin.py) tell the device path from a
 * fallback to the reference's CPU function: the replacement translation
 * units (the files under shim/) call the reference only when an entry returns non-zero,
 * and every such return is counted here. */
typedef struct gpcc_ctx_stats_t {
  int64_t calls_ok;          /* entries that returned GPCC_OK                 */
  int64_t calls_unsupported; /* returned GPCC_ERR_UNSUPPORTED (CPU keeps it)  */
  int64_t calls_failed;      /* returned any other error                      */
  int64_t points_ok;         /* points processed by the successful entries    */
} gpcc_ctx_stats_t;
int gpcc_ctx_stats(const gpcc_ctx* ctx, gpcc_ctx_stats_t* out);

/* The predicting encoder with direct predictors iterates its reconstruction pass and the rate
 * model's trajectory to their fixed point (see gpcc_pred_forward); a slice that has not settled
 * within the pass limit (64) is declined.  out[0] slices coded that way since the context was
 * created, out[1] passes over all of them, out[2] the most passes one slice took, out[3] slices
 * declined at the limit (none observed so far: bench.py's predicting leg reports these). */
int gpcc_ctx_pred_pass_stats(const gpcc_ctx* ctx, int64_t out[4]);

/* ------------------------------------------------------------------ */
/* host tier: one slice, host buffers, synchronous                      */

/* Replaces pcc::regionAdaptiveHierarchicalTransform (RAHT.h:47-57,
 * RAHT.cpp:1997-2018) for intra slices.
 *   morton  [n]     ascending Morton codes (mortonAddr, PCCMath.h:606)
 *   qp_off  [n][2]  per-point region QP offsets (QpSet::regionQpOffset),
 *                   NULL = all zero
 *   attrs   [n*c]   in: source attributes, row-major; out: reconstruction
 *                   (unclipped, RAHT.cpp:1969-1975)
 *   coeffs  [c*n]   out: quantised coefficients, planar (coeff[k*n+i],
 *                   RAHT.cpp:992-996), traversal order
 */
int gpcc_raht_forward(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int64_t* morton,
  const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int32_t n,
  int32_t c);

/* Replaces pcc::regionAdaptiveHierarchicalInverseTransform (RAHT.h:59-69,
 * RAHT.cpp:2037-2058).  coeffs is read, attrs [n*c] is written. */
int gpcc_raht_inverse(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int64_t* morton,
  const int32_t* qp_off, int32_t* attrs, const int32_t* coeffs, int32_t n,
  int32_t c);

/* RAHT with attribute inter prediction: replaces pcc::regionAdaptiveHierarchicalTransform /
 * ...InverseTransform (RAHT.h:47-69) called with attrInterPredParams.enableAttrInterPred and a reference
 * frame in paramsForInterRAHT (PCCTMC3Common.h:236-298; RAHT.cpp:849-972 estimate_layer_filter,
 * :1165-1198 the two trees in lock step, :1256-1347 per-layer decision, filter taps and block matching,
 * :1504-1549 the frame's block as prediction, :1810-1829 the decision).
 *   morton_ref [n_ref], attrs_ref [n_ref*c]  the reference frame, Morton order (paramsForInterRAHT.
 *                                            mortonCode / attributes)
 *   layer_modes [32], num_modes   attr_layer_code_mode: written by the encoder, read by the decoder
 *   filter_taps [32], num_taps    FilterTaps (quantised): written by the encoder when
 *                                 enable_filter_estimation, read by the decoder
 *                                 (the decoder accepts NULL for an array whose count is 0)
 * Everything else as gpcc_raht_forward / _inverse (qp_off: region QP offsets per point, NULL = none).  On the
 * device: every parameter
 * set, with sub-node prediction (the dependency kernels, the encoder's two candidates of a level as two
 * launches) and with the integer Haar kernel (the frame gets level arrays of its own) -- except, under the
 * Haar kernel, a frame whose tree height differs from the current one's by a non-multiple of three bits
 * (GPCC_ERR_UNSUPPORTED: the CPU keeps the slice).  The
 * per-layer decision compares two sums of doubles the reference accumulates in coding order with log2 of
 * the HOST's libm inside: the library fills its log2 table from the same libm and adds in the same order
 * (csrc/raht_inter.hpp); a coefficient magnitude beyond the table (2^20) returns GPCC_ERR_UNSUPPORTED. */
typedef struct gpcc_raht_inter_params {
  int32_t raht_inter_prediction_depth_minus1;
  int32_t raht_enable_inter_intra_layer_rdo;
  int32_t enable_filter_estimation;
  int32_t skip_init_layers_for_filtering;
} gpcc_raht_inter_params;

int gpcc_raht_forward_inter(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_raht_inter_params* inter,
  const int64_t* morton, const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c,
  const int64_t* morton_ref, const int32_t* attrs_ref, int32_t n_ref,
  int32_t* layer_modes, int32_t* num_modes, int32_t* filter_taps, int32_t* num_taps);

int gpcc_raht_inverse_inter(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_raht_inter_params* inter,
  const int64_t* morton, const int32_t* qp_off, int32_t* attrs, const int32_t* coeffs, int32_t n, int32_t c,
  const int64_t* morton_ref, const int32_t* attrs_ref, int32_t n_ref,
  const int32_t* layer_modes, int32_t num_modes, const int32_t* filter_taps, int32_t num_taps);

/* Replaces the Morton-code + std::sort(MortonCodeWithIndex) prologue of
 * encode/decode{Colors,Reflectances}TransformRaht
 * (AttributeEncoder.cpp:1225-1229,1316-1321; AttributeDecoder.cpp:538-542,
 * 624-628; ordering MortonCodeWithIndex::operator< PCCTMC3Common.h:184-190:
 * by code, ties by original index).
 *   xyz     [n][3]  point positions (Vec3<int32_t>, non-negative, < 2^21)
 *   morton  [n]     out: sorted codes
 *   order   [n]     out: original index of the i-th sorted point
 */
int gpcc_attr_morton_sort(
  gpcc_ctx* ctx, const int32_t* xyz, int32_t n, int64_t* morton,
  int32_t* order);

/* ------------------------------------------------------------------ */
/* device tier: batched over slices, buffers resident in HBM            */

/* A batch of `num_slices` independent slices laid out back to back.
 * slice s owns points [offsets[s], offsets[s+1]).  offsets is a HOST
 * array of num_slices+1 entries (offsets[0] == 0).  All d_* pointers are
 * device addresses (passed as void* so that no HIP header is needed).
 * For slice s with n_s points and base b = offsets[s]:
 *   d_morton  + b           int64  [n_s]
 *   d_qp_off  + 2*b         int32  [n_s][2]  (d_qp_off may be NULL)
 *   d_attrs   + c*b         int32  [n_s][c]
 *   d_coeffs  + c*b         int32  [c][n_s]   planar per slice
 * The call enqueues work on the context's stream and returns; results
 * are valid after gpcc_ctx_synchronize (or stream ordering). */
int gpcc_dev_raht_forward(
  gpcc_ctx* ctx, const gpcc_raht_params* params, int32_t num_slices,
  const int64_t* offsets, const void* d_morton, const void* d_qp_off,
  void* d_attrs, void* d_coeffs, int32_t c);

int gpcc_dev_raht_inverse(
  gpcc_ctx* ctx, const gpcc_raht_params* params, int32_t num_slices,
  const int64_t* offsets, const void* d_morton, const void* d_qp_off,
  void* d_attrs, const void* d_coeffs, int32_t c);

/* Morton encode + stable sort per slice.  d_xyz int32 [n][3];
 * d_morton int64 [n]; d_order int32 [n] (index local to the slice). */
int gpcc_dev_attr_morton_sort(
  gpcc_ctx* ctx, int32_t num_slices, const int64_t* offsets,
  const void* d_xyz, void* d_morton, void* d_order);

/* ------------------------------------------------------------------ */
/* lifting transform (predictors given)                                 */

#define GPCC_MAX_LODS 32

/* What encode/decode{Colors,Reflectances}Lift (AttributeEncoder.cpp:1379-1648,
 * AttributeDecoder.cpp:678-857) read besides the LoD structure: the
 * cumulative LoD sizes (AttributeLods::numPointsInLod, coarse to fine), the
 * QpSet (fixed_point_qp_offset = 24 for lifting, quantization.cpp:155-158),
 * the bit depth for the final clip and the last-component-prediction flag. */
typedef struct gpcc_lift_params {
  int32_t num_lods;
  int32_t num_points_in_lod[GPCC_MAX_LODS];
  int32_t last_component_prediction_enabled_flag;
  int32_t bitdepth;
  int32_t num_qp_layers;
  int32_t layer_qp[GPCC_MAX_QP_LAYERS][2];
  int32_t max_qp;
  int32_t fixed_point_qp_offset;
  /* AttributeParameterSet::scalable_lifting_enabled_flag: the quantisation
   * weight of a predictor is then a function of its level of detail alone
   * (computeQuantizationWeightsScalable, PCCTMC3Common.h:858-891, whole
   * slices: minGeomNodeSizeLog2 = 0) */
  int32_t scalable_lifting_enabled_flag;
  /* QP regions of the slice (AttributeBrickHeader::qpRegions -> QpSet::regions,
   * tmc3/quantization.cpp:100-117), for the entries that build the LoD structure themselves
   * (gpcc_*_encode_attr / _decode_attr, gpcc_dev_*, gpcc_multi_*): every point's offset is derived
   * from its position on the device as QpSet::regionQpOffset does (:195-204: the first region that
   * contains it, bounds inclusive).  gpcc_lift_forward / gpcc_pred_forward and their inverses take the
   * offsets as an array instead and ignore these. */
  int32_t num_qp_regions; /* 0 .. GPCC_MAX_QP_REGIONS */
  int32_t qp_region_min[GPCC_MAX_QP_REGIONS][3];
  int32_t qp_region_max[GPCC_MAX_QP_REGIONS][3];
  int32_t qp_region_offset[GPCC_MAX_QP_REGIONS][2];
} gpcc_lift_params;

/* The predictors of AttributeLods (AttributeCommon.h:89-94) as flat arrays in
 * PREDICTOR order (coding order, coarsest LoD first), after
 * PCCPredictor::computeWeights:
 *   neigh_count [n]     PCCPredictor::neighborCount (0..3)
 *   neigh_index [n][3]  PCCNeighborInfo::predictorIndex (< start of the LoD)
 *   neigh_weight[n][3]  PCCNeighborInfo::weight (8-bit fixed point, sum 256)
 *   indexes     [n]     AttributeLods::indexes: predictor order -> point index
 *   qp_off      [n][2]  region QP offset of each POINT (QpSet::regionQpOffset),
 *                       NULL = none
 * Forward: replaces the body of encodeColorsLift / encodeReflectancesLift
 * minus the entropy calls.  attrs [n][c] in point order: in source, out the
 * reconstructed (rounded, clipped) attributes, as the reference writes them
 * back with setColor/setReflectance.  coeffs [n][c] out: the quantised
 * values of each predictor in coding order (`values[]`, the input of
 * PCCResidualsEncoder::encode and of the zero-run counter).  lcp_coeffs
 * [GPCC_MAX_LODS] out: AttributeBrickHeader::attrLcpCoeffs (c == 3 and the
 * flag set, otherwise untouched). */
int gpcc_lift_forward(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, int32_t* coeffs, int8_t* lcp_coeffs);

/* Replaces decodeColorsLift / decodeReflectancesLift after the entropy
 * decode: coeffs and lcp_coeffs in, attrs [n][c] (point order) out. */
int gpcc_lift_inverse(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, const int32_t* coeffs, const int8_t* lcp_coeffs);

/* Flattened LoD-generation parameters: the AttributeParameterSet fields
 * buildPredictorsFast reads (hls.h:782-876) plus
 * AttributeBrickHeader::attr_dist2_delta: the parameter block of
 * gpcc_lod_build below. */
typedef struct gpcc_lod_params {
  int32_t attr_encoding;           /* 1 predicting, 2 lifting */
  int32_t lod_decimation_type;     /* LodDecimationMethod */
  int32_t num_detail_levels_minus1;
  int32_t num_pred_nearest_neighbours_minus1;
  int32_t intra_lod_search_range;
  int32_t inter_lod_search_range;
  int32_t prediction_with_distribution_enabled;
  int32_t lod_neigh_bias[3];
  int32_t intra_lod_prediction_skip_layers;
  int32_t dist2;
  int32_t attr_dist2_delta;
  int32_t canonical_point_order_flag;
  int32_t max_points_per_sort_log2_plus1;
  int32_t scalable_lifting_enabled_flag;
  int32_t max_neigh_range_minus1;
  int32_t pred_weight_blending_enabled_flag;
  int32_t lod_sampling_period[GPCC_MAX_LODS];
} gpcc_lod_params;

/* Replaces pcc::AttributeLods::generate (AttributeCommon.h:80-87,
 * AttributeCommon.cpp:44-72 -> buildPredictorsFast PCCTMC3Common.h:2300-2469
 * + PCCPredictor::computeWeights) for one intra slice:
 *   xyz [n][3] point positions (point order, as PCCPointSet3::positions)
 * out, all in PREDICTOR (coding) order, coarsest level of detail first:
 *   neigh_count [n], neigh_index [n][3] (predictor indices), neigh_weight
 *   [n][3] (8-bit weights), indexes [n] (predictor -> point index),
 *   num_points_in_lod [GPCC_MAX_LODS] cumulative, *num_lods.
 * All three decimators (lod_decimation_type 0 distance, 1 periodic, 2
 * centroid) and, for the predicting transform, blendWeights run on the
 * device.  canonical_point_order_flag / max_points_per_sort_log2_plus1
 * (PCCTMC3Common.h:2322-2331: the points taken as they come, or sorted in
 * chunks) are accepted when the points ARE in Morton order -- what the octree
 * geometry coder hands over; then neither changes the result.  Points in any
 * other order with those flags and inter prediction return
 * GPCC_ERR_UNSUPPORTED (the shim keeps them on the reference path).
 * scalable_lifting_enabled_flag (whole slices: minGeomNodeSizeLog2 = 0, no
 * points skipped by a partial decode) builds the structure of
 * PCCTMC3Common.h:2377-2448: always 21 levels, octree sub-sampling by LoD
 * index, node-corner positions in the search, neighbours beyond
 * max_neigh_range_minus1 dropped, the finer layers searched again while a new
 * layer outweighs them; num_detail_levels_minus1, lod_decimation_type, dist2
 * and the sampling periods are then not read.  neigh_weight entries at and
 * beyond neigh_count are unspecified then (the reference keeps the raw squared
 * distance of a dropped neighbour in its slot; nothing reads it). */
int gpcc_lod_build(
  gpcc_ctx* ctx, const gpcc_lod_params* params, const int32_t* xyz, int32_t n,
  int32_t* neigh_count, int32_t* neigh_index, int32_t* neigh_weight,
  int32_t* indexes, int32_t* num_points_in_lod, int32_t* num_lods);

/* AttributeLods::generate with attribute INTER prediction
 * (AttributeInterPredParams::enableAttrInterPred; buildPredictorsFast with
 * interRef, PCCTMC3Common.h:2348-2376, 2396-2400; the reference-frame part of
 * computeNearestNeighbors :1270-1292, :1606-1796; updatePredictors :2286-2293;
 * blendWeights :654-656): the neighbour search also takes candidates from the
 * reference frame xyz_ref [n_ref][3] (AttributeInterPredParams::
 * referencePointCloud, point order), search_range =
 * AttributeBrickHeader::attrInterPredSearchRange (it replaces both LoD search
 * ranges of the block), frame_distance = AttributeInterPredParams::
 * frameDistance.  Outputs as gpcc_lod_build, plus inter_ref [n][3]
 * (PCCNeighborInfo::interFrameRef): for a neighbour with the flag set
 * neigh_index is a POINT index of the reference frame (PCCNeighborInfo::
 * pointIndex) and its distance carried the frame distance into the weights.
 * Not combined with scalable lifting / canonical point order
 * (GPCC_ERR_UNSUPPORTED).
 * STATUS (round 3): bit-exact against the oracle on the MI355X
 * (tests/test_zz_gpu_inter_lod.py) and under the CPU wavefront emulator
 * (tests/test_emu_lod.py).  Consumers: gpcc_lift_forward_inter /
 * gpcc_pred_forward_inter (and the inverses) below; the shims call it for
 * slices with attribute inter prediction (seam 2: AttributeLods::generate,
 * seam 3: the operator factories). */
int gpcc_lod_build_inter(
  gpcc_ctx* ctx, const gpcc_lod_params* params, const int32_t* xyz, int32_t n,
  const int32_t* xyz_ref, int32_t n_ref, int32_t search_range,
  int32_t frame_distance, int32_t* neigh_count, int32_t* neigh_index,
  int32_t* neigh_weight, int32_t* indexes, int32_t* num_points_in_lod,
  int32_t* num_lods, int32_t* inter_ref);

/* Reflectance lifting over such a structure (encodeReflectancesLift /
 * decodeReflectancesLift with enableAttrInterPred, AttributeEncoder.cpp:
 * 1543-1648, AttributeDecoder.cpp:780-857; one component -- the reference's
 * colour driver takes no reference frame): predictors and inter_ref as
 * gpcc_lod_build_inter returns them, attrs_ref [n_ref] the reference frame's
 * reflectances in ITS point order.  PCCLiftPredict takes the frame's value
 * for a flagged neighbour; PCCLiftUpdate and the quantisation weights leave
 * it out (PCCTMC3Common.h:735-740, :799-800, :845-846).  Buffers otherwise
 * as gpcc_lift_forward / gpcc_lift_inverse with c = 1.
 * STATUS (round 3): the kernels are the intra ones (the frame's values sit
 * behind the n working values and flagged neighbours point there); that
 * arrangement runs under the CPU emulator against the oracle
 * (tests/test_emu_lod.py) and the entry's first hardware run is
 * tests/test_zz_gpu_inter_lod.py in the round-end tier. */
int gpcc_lift_forward_inter(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* inter_ref, const int32_t* indexes,
  int32_t* attrs, const int32_t* attrs_ref, int32_t n_ref, int32_t* coeffs);
int gpcc_lift_inverse_inter(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* inter_ref, const int32_t* indexes,
  int32_t* attrs, const int32_t* attrs_ref, int32_t n_ref,
  const int32_t* coeffs);

/* PCCPredictor::computeWeights (PCCTMC3Common.h:589-633) for n predictors:
 * squared distances in neigh_weight (uint64 [n][3]) -> 8-bit weights
 * (int32 [n][3]); neigh_count is updated in place (far neighbours are
 * dropped). */
int gpcc_lod_compute_weights(
  gpcc_ctx* ctx, int32_t n, int32_t* neigh_count, const uint64_t* dist2,
  int32_t* neigh_weight);

/* Per-kernel timing of the most recent gpcc_dev_raht_* call on this
 * context, measured with HIP events on the context's stream when
 * profiling is enabled (gpcc_ctx_set_profiling(ctx, 1)).  Returns the
 * number of entries written (<= max_entries).  names[i] points to a
 * static string. */
typedef struct gpcc_kernel_time {
  const char* name;
  double total_ms;
  int32_t launches;
} gpcc_kernel_time;
int gpcc_ctx_set_profiling(gpcc_ctx* ctx, int enable);
int gpcc_ctx_kernel_times(
  gpcc_ctx* ctx, gpcc_kernel_time* out, int32_t max_entries);

/* The whole RAHT driver of one slice minus the entropy loop --
 * encodeColorsTransformRaht / encodeReflectancesTransformRaht
 * (tmc3/AttributeEncoder.cpp:1306-1375 / 1214-1302) and
 * decodeColorsRaht / decodeReflectancesRaht (tmc3/AttributeDecoder.cpp:613-674
 * / 527-609): Morton codes of xyz[n][3], sort by (code, index), gather the
 * attributes into that order, transform, clip the reconstruction to
 * [0, 2^bitdepth - 1] and scatter it back by original point index -- one
 * upload, one download, everything in between on the device.
 *   attrs  [n][c] in POINT order: encode in: source, out: clipped
 *          reconstruction; decode out: clipped reconstruction
 *   coeffs planar [c][n] in Morton order, exactly what the entropy loop reads
 * Region QP offsets (QpSet::regionQpOffset) are taken as zero, as in every CTC
 * configuration; use gpcc_attr_morton_sort + gpcc_raht_forward for regions (the LoD-based
 * one-call entries below take the regions in their parameter blocks). */
int gpcc_raht_encode_attr(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz,
  int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth);
int gpcc_raht_decode_attr(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz,
  int32_t* attrs, const int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth);

/* The lifting attribute coder of one slice minus the entropy loop: what
 * AttributeEncoder::encode does for AttributeEncoding::kLiftingTransform --
 * AttributeLods::generate (AttributeEncoder.cpp:575-579) followed by
 * encodeColorsLift / encodeReflectancesLift (:1379-1648) -- resp.
 * AttributeDecoder::decode (AttributeDecoder.cpp:292-296, 678-857), in one
 * call: the LoD structure is built and consumed on the device and never
 * crosses PCIe.
 *   lod     LoD parameters (as for gpcc_lod_build)
 *   lift    in: QP layers, bit depth, last_component_prediction flag;
 *           out: num_lods / num_points_in_lod of the structure that was built
 *   xyz     [n][3] positions, point order
 *   attrs   [n][c] point order; encode in: source, out: clipped
 *           reconstruction; decode out: clipped reconstruction
 *   coeffs  [n][c] coding order (what the entropy loop codes / decoded)
 *   lcp_coeffs [GPCC_MAX_LODS] last-component prediction coefficients
 *   indexes [n] out, may be NULL: predictor -> point index (coding order)
 */
int gpcc_lift_encode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift,
  const int32_t* xyz, int32_t* attrs, int32_t* coeffs, int8_t* lcp_coeffs,
  int32_t* indexes, int32_t n, int32_t c);
int gpcc_lift_decode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift,
  const int32_t* xyz, int32_t* attrs, const int32_t* coeffs,
  const int8_t* lcp_coeffs, int32_t* indexes, int32_t n, int32_t c);

/* ------------------------------------------------------------------ */
/* predicting transform                                                  */

/* What encode/decode{Colors,Reflectances}Pred (AttributeEncoder.cpp:749-853,
 * 1075-1210, AttributeDecoder.cpp:328-523) read besides the LoD structure:
 * the cumulative LoD sizes, the QpSet (fixed_point_qp_offset = 0 for this
 * transform, quantization.cpp:151-158), the bit depth and the
 * AttributeParameterSet fields of the prediction-mode and inter-component
 * tools (hls.h:782-876). */
typedef struct gpcc_pred_params {
  int32_t num_lods;
  int32_t num_points_in_lod[GPCC_MAX_LODS];
  int32_t bitdepth;
  int32_t num_qp_layers;
  int32_t layer_qp[GPCC_MAX_QP_LAYERS][2];
  int32_t max_qp;
  int32_t max_num_direct_predictors;
  int32_t direct_avg_predictor_disabled_flag;
  int32_t adaptive_prediction_threshold; /* aps.adaptivePredictionThreshold(desc):
                                          * the APS value << max(0, bitdepth - 8) */
  int32_t inter_component_prediction_enabled_flag;
  int32_t quant_neigh_weight[3];
  int32_t max_num_detail_levels;         /* aps.maxNumDetailLevels(): icp_coeffs
                                          * beyond it are zero */
  int32_t scalable_lifting_enabled_flag; /* quantisation weights by level of detail
                                          * (computeQuantizationWeightsScalable,
                                          * PCCTMC3Common.h:858-891; whole slices) instead
                                          * of quant_neigh_weight */
  /* QP regions of the slice (AttributeBrickHeader::qpRegions -> QpSet::regions,
   * tmc3/quantization.cpp:100-117), for the entries that build the LoD structure themselves
   * (gpcc_*_encode_attr / _decode_attr, gpcc_dev_*, gpcc_multi_*): every point's offset is derived
   * from its position on the device as QpSet::regionQpOffset does (:195-204: the first region that
   * contains it, bounds inclusive).  gpcc_lift_forward / gpcc_pred_forward and their inverses take the
   * offsets as an array instead and ignore these. */
  int32_t num_qp_regions; /* 0 .. GPCC_MAX_QP_REGIONS */
  int32_t qp_region_min[GPCC_MAX_QP_REGIONS][3];
  int32_t qp_region_max[GPCC_MAX_QP_REGIONS][3];
  int32_t qp_region_offset[GPCC_MAX_QP_REGIONS][2];
} gpcc_pred_params;

/* Replaces decodeColorsPred / decodeReflectancesPred after the entropy decode
 * (AttributeDecoder.cpp:328-523), predictors as for gpcc_lift_forward:
 *   values [n][c] in: the decoded `values` of every predictor in coding order
 *          (zero runs expanded; the prediction mode still hidden in their
 *          parities, decodePredModeColor / decodePredModeRefl :288-323,
 *          :404-447)
 *   icp_coeffs [GPCC_MAX_LODS][3] in: AttributeBrickHeader::icpCoeffs (c == 3
 *          and the flag set; otherwise ignored, may be NULL)
 *   attrs  [n][c] out, point order: the reconstruction.
 * A point depends on the reconstruction of its neighbours -- also of its own
 * level of detail unless intra_lod_prediction_skip_layers excludes that: the
 * device walks the dependency DAG (pred_kernels.hpp); the result is the
 * reference's for every configuration, the time grows with the depth of the
 * DAG (a single level of detail on a scan-ordered LiDAR frame is one chain). */
int gpcc_pred_inverse(
  gpcc_ctx* ctx, const gpcc_pred_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, const int32_t* values, const int8_t* icp_coeffs);

/* The reflectance predicting transform over such a structure
 * (encodeReflectancesPred / decodeReflectancesPred with enableAttrInterPred,
 * AttributeEncoder.cpp:749-853, AttributeDecoder.cpp:328-400): every neighbour
 * value the coder reads -- the prediction, the eligibility test of the direct
 * predictors, their evaluation -- is the reference frame's reflectance for a
 * flagged neighbour, and such a neighbour takes no quantisation-weight share.
 * Arguments as gpcc_lift_forward_inter; values as gpcc_pred_forward / _inverse.
 * STATUS (round 3): the DAG pass in its inter build and the spare-entry
 * arrangement run under the CPU emulator against the oracle
 * (tests/test_emu_lod.py); first hardware run: tests/test_zz_gpu_inter_lod.py
 * in the round-end tier. */
int gpcc_pred_forward_inter(
  gpcc_ctx* ctx, const gpcc_pred_params* params, int32_t n,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* inter_ref, const int32_t* indexes,
  int32_t* attrs, const int32_t* attrs_ref, int32_t n_ref, int32_t* values);
int gpcc_pred_inverse_inter(
  gpcc_ctx* ctx, const gpcc_pred_params* params, int32_t n,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* inter_ref, const int32_t* indexes,
  int32_t* attrs, const int32_t* attrs_ref, int32_t n_ref,
  const int32_t* values);


/* Replaces the body of encodeColorsPred / encodeReflectancesPred minus the
 * entropy calls: attrs in source / out reconstruction, values [n][c] out (the
 * prediction mode in the low bits of the magnitudes, encodePredModeColor /
 * ...Refl), icp_coeffs out (computeInterComponentPredictionCoeffs,
 * AttributeEncoder.cpp:990-1071).  With direct predictors the encoder's mode
 * decision (decidePredModeColor :896-985, decidePredModeRefl :663-745) reads a
 * rate model that every earlier point has updated (:136-222): the device runs
 * the reconstruction pass with the model's states as an input, recomputes the
 * states from the pass's values, and repeats until a pass changes no value --
 * the sequential coder's result (a few passes per slice).  The estimate's log2
 * values come from a table filled by the HOST's libm, so the doubles are the
 * reference's.  A slice that has not settled after 64 passes returns
 * GPCC_ERR_UNSUPPORTED with attrs restored (not observed). */
int gpcc_pred_forward(
  gpcc_ctx* ctx, const gpcc_pred_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, int32_t* values, int8_t* icp_coeffs);

/* The predicting attribute coder of one slice minus the entropy loop, as
 * gpcc_lift_encode_attr / gpcc_lift_decode_attr: AttributeLods::generate
 * (with blendWeights when the APS asks for it) and the transform in one call,
 * the predictors never leave the device.  pred: in tools / QP; out num_lods,
 * num_points_in_lod of the structure that was built.  QP regions: the
 * qp_region_* fields of the parameter block (round 4). */
int gpcc_pred_encode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_pred_params* pred,
  const int32_t* xyz, int32_t* attrs, int32_t* values, int8_t* icp_coeffs,
  int32_t* indexes, int32_t n, int32_t c);
int gpcc_pred_decode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_pred_params* pred,
  const int32_t* xyz, int32_t* attrs, const int32_t* values,
  const int8_t* icp_coeffs, int32_t* indexes, int32_t n, int32_t c);

/* gpcc_raht_encode_attr whose result is the symbol stream of the entropy
 * loop (see gpcc_zero_run_pack) instead of the coefficient array: runs [n],
 * values [n][c] (only *num_symbols entries are written and copied), the
 * zero-run formation runs on the device where the coefficients are. */
int gpcc_raht_encode_attr_packed(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz,
  int32_t* attrs, int32_t* runs, int32_t* values, int32_t* num_symbols,
  int32_t* trailing_run, int32_t n, int32_t c, int32_t bitdepth);

/* QP regions of a slice (AttributeBrickHeader::qpRegions -> QpSet::regions, quantization.cpp:196-204): the
 * offset of the FIRST box that contains a point is added to its layer QPs.  Field names as in
 * gpcc_lift_params / gpcc_pred_params. */
typedef struct gpcc_qp_regions {
  int32_t num_qp_regions; /* 0 .. GPCC_MAX_QP_REGIONS */
  int32_t qp_region_min[GPCC_MAX_QP_REGIONS][3];
  int32_t qp_region_max[GPCC_MAX_QP_REGIONS][3];
  int32_t qp_region_offset[GPCC_MAX_QP_REGIONS][2];
} gpcc_qp_regions;

/* The RAHT slice drivers with QP regions (round 5): gpcc_raht_encode_attr_packed / gpcc_raht_decode_attr with the
 * per-point offsets the reference's drivers derive with qpSet.regionQpOffset (AttributeEncoder.cpp:1262, 1336;
 * AttributeDecoder.cpp:585, 648) computed on the device from the positions.  regions == NULL or
 * num_qp_regions == 0: exactly the entries without regions. */
int gpcc_raht_encode_attr_packed_regions(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_qp_regions* regions, const int32_t* xyz,
  int32_t* attrs, int32_t* runs, int32_t* values, int32_t* num_symbols,
  int32_t* trailing_run, int32_t n, int32_t c, int32_t bitdepth);
int gpcc_raht_decode_attr_regions(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_qp_regions* regions, const int32_t* xyz,
  int32_t* attrs, const int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth);

/* Zero-run formation of a coefficient stream: the part of the reference's
 * entropy loops that is not the arithmetic coder (AttributeEncoder.cpp:
 * 1279-1291 / 1347-1362 RAHT, 1458-1474 / 1617-1633 lifting).  A position
 * whose c values are all zero extends the current run; any other position
 * emits (run, values).  The caller then issues
 *   for k < *num_symbols: encodeRunLength(runs[k]); encode(values[k][0..c));
 *   if (*trailing_run) encodeRunLength(*trailing_run);
 * to the reference's PCCResidualsEncoder -- the bitstream is identical
 * (tests/test_symbols.py) and only the non-zero symbols cross PCIe.
 *   coeffs  planar != 0: [c][n] (RAHT);  planar == 0: [n][c] (lifting)
 *   runs    [n] out (num_symbols used), values [n][c] out
 * Host tier. */
int gpcc_zero_run_pack(
  gpcc_ctx* ctx, const int32_t* coeffs, int32_t n, int32_t c, int32_t planar,
  int32_t* runs, int32_t* values, int32_t* num_symbols, int32_t* trailing_run);

/* estimateDist2 (tmc3/AttributeEncoder.cpp:1684-1720, called from
 * tmc3/encoder.cpp:1203 to derive attr_dist2_delta): for every
 * sampling_period-th point of xyz[n][3] (coded order) the squared distance to
 * the nearest other point within +-search_range positions; the
 * `percentile`-th of those minima picks the smallest shift with
 * 3 << (2*shift) >= dist2.  The distance scan runs on the device, the
 * selection (std::nth_element in the reference) on the host.  Host tier. */
int gpcc_estimate_dist2(
  gpcc_ctx* ctx, const int32_t* xyz, int32_t n, int32_t sampling_period,
  int32_t search_range, float percentile, int32_t* shift_bits);

/* Binarisation of the residual symbols: the part of PCCResidualsEncoder
 * (AttributeEncoder.cpp:57-307 -- encodeRunLength :227-254, encodeSymbol
 * :259-272, encode :278-307, with the exp-Golomb binarisations of
 * entropyutils.h:142-183) that is not the adaptive arithmetic coder.  For the
 * symbol stream of gpcc_zero_run_pack / gpcc_raht_encode_attr_packed (runs,
 * values, trailing run) it returns the binary decisions in coding order, one
 * byte each: (context << 1) | bin with context = 0..4 ctxRunLen[5], 5..18
 * ctxCoeffGtN[2][7], 19..24 ctxCoeffRemPrefix[2][3], 25..30
 * ctxCoeffRemSuffix[2][3] (AttributeCommon.h:54-57), 31 = bypass.  Which
 * context a decision uses depends on the symbol alone, so all symbols are
 * binarised in parallel; the caller's loop is
 *   for (b : bins) ctx(b) == 31 ? enc.encode(bin(b)) : enc.encode(bin(b), model[ctx(b)]);
 * on the reference's own coder: the bitstream is identical (tests/test_bins.py).
 *   c = 3 (colour) or 1;  bins [cap] out;  *num_bins out (also when cap is too
 *   small: the call then fails with GPCC_ERR_INVALID_ARG).  Host tier. */
int gpcc_binarise_symbols(
  gpcc_ctx* ctx, const int32_t* runs, const int32_t* values,
  int32_t num_symbols, int32_t trailing_run, int32_t c, uint8_t* bins,
  int64_t cap, int64_t* num_bins);

/* ------------------------------------------------------------------ */
/* device tier of the LoD build and the lifting coder                   */
/* The same operations on buffers resident in HBM, num_slices slices back to
 * back (offsets[0] = 0 .. offsets[num_slices] = total points): nothing is
 * allocated per call (workspace from the context's arena) and nothing crosses
 * PCIe except, per level of detail, the size of the retained list the host
 * needs to size the next launches.  Positions are not inspected on the host
 * (coordinates must lie in [0, 2^21)); gpcc_ctx_set_morton_bits bounds the
 * radix passes of the Morton sort.  The slices are processed one after the
 * other on the context's stream; the calls return when the batch is done.
 *
 * gpcc_dev_lod_build: AttributeLods::generate for every slice.
 *   d_xyz [N][3]; out d_neigh_count [N], d_neigh_index [N][3], d_neigh_weight
 *   [N][3], d_indexes [N] -- each slice's in its own predictor order, indices
 *   slice relative; host: num_points_in_lod [num_slices][GPCC_MAX_LODS],
 *   num_lods [num_slices]. */
int gpcc_dev_lod_build(
  gpcc_ctx* ctx, const gpcc_lod_params* params, int32_t num_slices,
  const int64_t* offsets, const void* d_xyz, void* d_neigh_count,
  void* d_neigh_index, void* d_neigh_weight, void* d_indexes,
  int32_t* num_points_in_lod, int32_t* num_lods);

/* gpcc_dev_lift_encode_attr / _decode_attr: gpcc_lift_encode_attr /
 * gpcc_lift_decode_attr for every slice, attributes and coefficients in place
 * in the caller's device buffers.
 *   lift   [num_slices] parameter blocks (in: QP layers etc.; out: the LoD
 *          structure of each slice)
 *   d_attrs [N][c] point order; d_coeffs [N][c] coding order per slice
 *   lcp_coeffs host [num_slices][GPCC_MAX_LODS] (c == 3 and the flag set)
 *   d_indexes [N] out, may be NULL */
int gpcc_dev_lift_encode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift,
  int32_t num_slices, const int64_t* offsets, const void* d_xyz, void* d_attrs,
  void* d_coeffs, int8_t* lcp_coeffs, void* d_indexes, int32_t c);
int gpcc_dev_lift_decode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift,
  int32_t num_slices, const int64_t* offsets, const void* d_xyz, void* d_attrs,
  const void* d_coeffs, const int8_t* lcp_coeffs, void* d_indexes, int32_t c);

/* gpcc_dev_pred_encode_attr / _decode_attr: gpcc_pred_encode_attr /
 * gpcc_pred_decode_attr for every slice of a batch resident in HBM (slices
 * concurrent on the context's lanes, as for the lifting coder).
 *   pred   [num_slices] parameter blocks (in: tools, QP; out: the LoD structure)
 *   d_attrs [N][c] point order; d_values [N][c] coding order per slice
 *   icp_coeffs host [num_slices][GPCC_MAX_LODS][3] (c == 3 and the flag set)
 *   d_indexes [N] out, may be NULL.  The encoder declines direct predictors
 *   (GPCC_ERR_UNSUPPORTED), see gpcc_pred_forward. */
int gpcc_dev_pred_encode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_pred_params* pred,
  int32_t num_slices, const int64_t* offsets, const void* d_xyz, void* d_attrs,
  void* d_values, int8_t* icp_coeffs, void* d_indexes, int32_t c);
int gpcc_dev_pred_decode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_pred_params* pred,
  int32_t num_slices, const int64_t* offsets, const void* d_xyz, void* d_attrs,
  const void* d_values, const int8_t* icp_coeffs, void* d_indexes, int32_t c);

/* ------------------------------------------------------------------ */
/* several GPUs from one host process                                   */
/* Slices are the reference's independent units (tmc3/encoder.cpp:544-571): a
 * single-process C++ host (the reference is one) shards a batch of slices over
 * the GPUs of a node with these entries.  gpcc_multi owns one context per
 * listed device; a call gives every device a contiguous, size-balanced run of
 * the slices, runs the transforms concurrently and gathers reconstructions and
 * coefficients on devices[0] -- RCCL send / receive over xGMI (librccl is
 * loaded on first use) -- the COEFFICIENTS, which feed one arithmetic coder; the
 * reconstruction is downloaded from the device that made it.  The caller's
 * buffers are pinned for the duration of the call (hipHostRegister) so that the
 * uploads of all devices overlap.
 * Listing one physical device several times is allowed (the gather is then a
 * device copy): it exercises the sharding on a one-GPU box.
 *   offsets [num_slices + 1], morton [N] (ascending per slice), attrs [N][c]
 *   (forward: in source, out reconstruction; inverse: out), coeffs planar per
 *   slice as for gpcc_dev_raht_forward.  Region QP offsets are taken as zero. */
typedef struct gpcc_multi gpcc_multi;
int gpcc_multi_create(const int32_t* devices, int32_t num_devices, gpcc_multi** out);
void gpcc_multi_destroy(gpcc_multi* m);
int gpcc_multi_num_devices(const gpcc_multi* m);
int gpcc_multi_uses_rccl(const gpcc_multi* m); /* 1: the gather goes through RCCL */
/* librccl loads, its symbols resolve and a one-rank communicator moves a buffer
 * through ncclSend / ncclRecv on `device` (what a single-GPU box can check of
 * the gather's transport); 0 = fine. */
int gpcc_multi_rccl_selftest(int32_t device);
int gpcc_multi_raht_forward(
  gpcc_multi* m, const gpcc_raht_params* params, int32_t num_slices,
  const int64_t* offsets, const int64_t* morton, int32_t* attrs,
  int32_t* coeffs, int32_t c);
int gpcc_multi_raht_inverse(
  gpcc_multi* m, const gpcc_raht_params* params, int32_t num_slices,
  const int64_t* offsets, const int64_t* morton, int32_t* attrs,
  const int32_t* coeffs, int32_t c);

/* The LoD-based coders on several GPUs: gpcc_lift_encode_attr /
 * gpcc_lift_decode_attr / gpcc_pred_encode_attr / gpcc_pred_decode_attr for every
 * slice of a batch (BASELINE configs[2]: five slices), the slices sharded over
 * the devices as above, each device driven by its own host thread for the
 * duration of the call.  Host buffers: xyz [N][3], attrs [N][c] point order per
 * slice, coeffs / values [N][c] coding order per slice (they feed the host's
 * arithmetic coder, so nothing is gathered between devices), lift / pred
 * [num_slices] parameter blocks (in: QP, tools; out: each slice's LoD structure),
 * lcp_coeffs [num_slices][GPCC_MAX_LODS], icp_coeffs [num_slices][GPCC_MAX_LODS][3],
 * indexes [N] out, may be NULL.  The first failing slice fails the call. */
int gpcc_multi_lift_encode_attr(
  gpcc_multi* m, const gpcc_lod_params* lod, gpcc_lift_params* lift,
  int32_t num_slices, const int64_t* offsets, const int32_t* xyz, int32_t* attrs,
  int32_t* coeffs, int8_t* lcp_coeffs, int32_t* indexes, int32_t c);
int gpcc_multi_lift_decode_attr(
  gpcc_multi* m, const gpcc_lod_params* lod, gpcc_lift_params* lift,
  int32_t num_slices, const int64_t* offsets, const int32_t* xyz, int32_t* attrs,
  const int32_t* coeffs, const int8_t* lcp_coeffs, int32_t* indexes, int32_t c);
int gpcc_multi_pred_encode_attr(
  gpcc_multi* m, const gpcc_lod_params* lod, gpcc_pred_params* pred,
  int32_t num_slices, const int64_t* offsets, const int32_t* xyz, int32_t* attrs,
  int32_t* values, int8_t* icp_coeffs, int32_t* indexes, int32_t c);
int gpcc_multi_pred_decode_attr(
  gpcc_multi* m, const gpcc_lod_params* lod, gpcc_pred_params* pred,
  int32_t num_slices, const int64_t* offsets, const int32_t* xyz, int32_t* attrs,
  const int32_t* values, const int8_t* icp_coeffs, int32_t* indexes, int32_t c);

/* ------------------------------------------------------------------ */
/* attribute transfer ("recolouring") onto a re-quantised geometry       */

/* RecolourParams (tmc3/pointset_processing.h:47-63) + the attribute's bit depth
 * (AttributeDescription::bitdepth, hls.h:286-299).  The reference's defaults
 * (TMC3.cpp:1501-1550): search_range 1, 8 / 1 neighbours, distance-weighted
 * averages, dist offsets 4, every max_*_dist2 1000 (>= 512 means "no limit"),
 * skip_avg_if_identical_fwd 1, _bwd 0. */
typedef struct gpcc_recolour_params {
  double dist_offset_fwd, dist_offset_bwd;
  double max_geometry_dist2_fwd, max_geometry_dist2_bwd;
  double max_attribute_dist2_fwd, max_attribute_dist2_bwd;
  int32_t search_range;
  int32_t num_neighbours_fwd, num_neighbours_bwd; /* 1 .. 8 each */
  int32_t use_dist_weighted_avg_fwd, use_dist_weighted_avg_bwd;
  int32_t skip_avg_if_identical_fwd, skip_avg_if_identical_bwd;
  int32_t bitdepth;
} gpcc_recolour_params;

/* Replaces pcc::recolour (pointset_processing.h:138-144,
 * pointset_processing.cpp:926-957 -> recolourColour :253-594 for c == 3,
 * recolourReflectance :618-916 for c == 1), called by the encoder after geometry
 * quantisation (tmc3/encoder.cpp: the attributes of the source cloud are
 * transferred to the points of the coded geometry).
 *   src_xyz [ns][3], src_attrs [ns][c]   the source cloud (values 0 .. 65535)
 *   tgt_xyz [nt][3]                      the target positions
 *   tgt_attrs [nt][c]                    out
 *   source_to_target_scale, target_to_source_offset[3]:
 *     posInTgt = posInSrc * scale - offset (the reference passes the scale as float)
 * Forward: the K nearest source points of every target point (exact, squared
 * distances in double exactly as nanoflann's L2 adaptor sums them); backward:
 * every source point is appended to the list of its nearest target points; then
 * the blend and the +-search_range refinement of pointset_processing.cpp.
 * PARITY: identical to the reference, equidistant candidates included.  All arithmetic is the
 * reference's (double, same order); where candidates are at EQUAL distance -- on a voxelised cloud
 * at a dyadic scale: at nearly every point -- the reference's result is the order its containers
 * produce, so they are rebuilt on the device: nanoflann's k-d trees (leaf size 10: divideTree /
 * middleSplit_ / planeSplit, dependencies/nanoflann/nanoflann.hpp:872-998) level by level, its search
 * (searchLevel :1308, KNNResultSet::addPoint :175) as a stack walk that visits candidates in
 * nanoflann's order, and the backward lists in source order sorted by libstdc++'s std::sort
 * (introsort, not stable beyond 16 entries).  oracle/recolour_oracle.c restates the same and is
 * pinned to the compiled reference (tests/test_oracle_recolour.py); the device equals both
 * (tests/test_gpu_recolour.py).  Needs ns >= num_neighbours_fwd and
 * nt >= num_neighbours_bwd (else GPCC_ERR_UNSUPPORTED).  A finite
 * max_geometry_dist2_fwd (< 512) is reproduced as the reference behaves (round 5): its result
 * vectors live outside its loop and its test looks at the farthest neighbour FOUND (:292-313), so
 * the first target whose k-th neighbour lies beyond the limit shrinks them to one entry for itself
 * and for every LATER target -- those take the colour of their nearest source point.  A k-d tree
 * deeper than 64 levels is declined (the search's stack).  Host tier. */
int gpcc_recolour(
  gpcc_ctx* ctx, const gpcc_recolour_params* params, const int32_t* src_xyz,
  const int32_t* src_attrs, int32_t ns, const int32_t* tgt_xyz, int32_t nt,
  int32_t c, float source_to_target_scale, const int32_t target_to_source_offset[3],
  int32_t* tgt_attrs);

/* ------------------------------------------------------------------ */
/* debug / test support (exported by every build; nothing an integrator */
/* needs): used by tests/ and tools/ only                               */
/* ------------------------------------------------------------------ */

/* Allocation events of the context since it was created: out[0] arena (re)allocations, out[1]
 * misses of the pooled device buffers, out[2] / out[3] (re)allocations of the pinned staging of
 * the compact level pass / of the level kernels.  A steady loop shows none. */
int gpcc_debug_alloc_events(const gpcc_ctx* ctx, long long out[4]);
/* 1 when the library was built with -DGPCC_EXPERIMENTS=1 (opt-in kernel variants compiled in). */
int gpcc_debug_has_experiments(void);
/* Guard bands compared so far in this process (0 unless GPCC_GUARD=1 in the environment). */
unsigned long long gpcc_debug_guard_checks(void);
/* The inter-frame encoder's rate sum (csrc/raht_inter.hpp, rate_sum_kernel) on its own:
 * out[e] = terms[e][0] + terms[e][1] + ... for the two estimates e, doubles added in order. */
int gpcc_debug_rate_sum(gpcc_ctx* ctx, const double* terms, int32_t count, double out[2]);
/* With GPCC_GUARD=1: writes 16 bytes past a pooled block (mode 0) or an arena sub-allocation
 * (mode 1) ON PURPOSE -- the process must stop with the guard-band message; without guard mode
 * it does nothing.  tests/test_gpu_guard.py runs it in a child process. */
int gpcc_debug_guard_selftest(gpcc_ctx* ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif /* GPCC_ATTR_MI355_H */
