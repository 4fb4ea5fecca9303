"""Flattened RAHT parameter block (ctypes mirror of ``gpcc_raht_params`` in
include/gpcc_attr_mi355.h) and the presets the reference encoder hands to
``regionAdaptiveHierarchicalTransform`` after its own sanitising
(reference: tmc3/TMC3.cpp:1270-1345,1883-1912, tmc3/encoder.cpp:710-842,
cfg/octree-raht-ctc-*.yaml)."""
import ctypes as C

GPCC_MAX_QP_LAYERS = 32
GPCC_MAX_QP_REGIONS = 8
GPCC_MAX_AC_QP_LAYERS = 32


class RahtInterParams(C.Structure):
    """gpcc_raht_inter_params: the tools of attribute inter prediction in RAHT
    (AttributeInterPredParamsForRAHT, PCCTMC3Common.h:236-250)"""
    _fields_ = [
        ("raht_inter_prediction_depth_minus1", C.c_int32),
        ("raht_enable_inter_intra_layer_rdo", C.c_int32),
        ("enable_filter_estimation", C.c_int32),
        ("skip_init_layers_for_filtering", C.c_int32),
    ]


class RahtParams(C.Structure):
    _fields_ = [
        ("raht_prediction_enabled_flag", C.c_int32),
        ("integer_haar_enable_flag", C.c_int32),
        ("raht_prediction_threshold0", C.c_int32),
        ("raht_prediction_threshold1", C.c_int32),
        ("raht_subnode_prediction_enabled_flag", C.c_int32),
        ("raht_prediction_search_range", C.c_int32),
        ("pred_weight_parent", C.c_int32 * 19),
        ("pred_weight_child", C.c_int32 * 12),
        ("raht_extension", C.c_int32),
        ("num_qp_layers", C.c_int32),
        ("layer_qp", (C.c_int32 * 2) * GPCC_MAX_QP_LAYERS),
        ("max_qp", C.c_int32),
        ("fixed_point_qp_offset", C.c_int32),
        ("num_ac_qp_layers", C.c_int32),
        ("ac_qp_offset", ((C.c_int32 * 2) * 7) * GPCC_MAX_AC_QP_LAYERS),
    ]

    def set_prediction_weights(self, w):
        """RahtPredictionParams::setPredictionWeights (hls.h:456-465)."""
        w = list(w)
        child = [w[4], w[4], w[3], w[4], w[3], w[3], w[4], w[4], w[4], w[4], w[4], w[4]]
        parent = [w[0], w[1], w[1], w[1], w[2], w[2], w[2], w[2], w[2], w[1],
                  w[2], w[1], w[1], w[2], w[2], w[2], w[2], w[2], w[2]]
        for i, v in enumerate(parent):
            self.pred_weight_parent[i] = v
        for i, v in enumerate(child):
            self.pred_weight_child[i] = v

    def set_layers(self, layers):
        """QpSet::layers: list of (luma qp, chroma qp offset)."""
        assert 1 <= len(layers) <= GPCC_MAX_QP_LAYERS
        self.num_qp_layers = len(layers)
        for i, (a, b) in enumerate(layers):
            self.layer_qp[i][0] = a
            self.layer_qp[i][1] = b

    def set_ac_offsets(self, layers):
        """QpSet::rahtAcCoeffQps: list of 7 (luma, chroma) pairs per layer."""
        assert len(layers) <= GPCC_MAX_AC_QP_LAYERS
        self.num_ac_qp_layers = len(layers)
        for i, lay in enumerate(layers):
            for j, (a, b) in enumerate(lay):
                self.ac_qp_offset[i][j][0] = a
                self.ac_qp_offset[i][j][1] = b

    def copy(self):
        o = RahtParams()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(RahtParams))
        return o


def raht_params(qp=34, chroma_offset=-1, bitdepth=8, prediction=True,
                subnode=True, haar=False, extension=True, search_range=50000,
                threshold0=2, threshold1=6, weights=(9, 3, 1, 5, 2),
                layers=None, ac_offsets=None):
    """The effective values of the CTC RAHT configs:
    cfg/octree-raht-ctc-lossless-geom-lossy-attrs.yaml (qp 22..46,
    qpChromaOffset -1, search range 50000 / 2500 for cat3-frame) and, with
    ``haar=True, qp=4``, ...-lossless-attrs.yaml."""
    p = RahtParams()
    p.raht_prediction_enabled_flag = int(prediction)
    p.integer_haar_enable_flag = int(haar)
    p.raht_prediction_threshold0 = threshold0
    p.raht_prediction_threshold1 = threshold1
    p.raht_subnode_prediction_enabled_flag = int(subnode)
    p.raht_prediction_search_range = search_range
    p.set_prediction_weights(weights)
    p.raht_extension = int(extension)
    p.set_layers(layers if layers is not None else [(qp, chroma_offset)])
    p.max_qp = 51 + 6 * (bitdepth - 8)   # quantization.cpp:151
    p.fixed_point_qp_offset = 0          # RAHT: quantization.cpp:155-158
    p.set_ac_offsets(ac_offsets or [])
    return p


GPCC_MAX_LODS = 32


class LiftParams(C.Structure):
    """ctypes mirror of gpcc_lift_params."""
    _fields_ = [
        ("num_lods", C.c_int32),
        ("num_points_in_lod", C.c_int32 * GPCC_MAX_LODS),
        ("last_component_prediction_enabled_flag", C.c_int32),
        ("bitdepth", C.c_int32),
        ("num_qp_layers", C.c_int32),
        ("layer_qp", (C.c_int32 * 2) * GPCC_MAX_QP_LAYERS),
        ("max_qp", C.c_int32),
        ("fixed_point_qp_offset", C.c_int32),
        ("scalable_lifting_enabled_flag", C.c_int32),
        ("num_qp_regions", C.c_int32),
        ("qp_region_min", (C.c_int32 * 3) * GPCC_MAX_QP_REGIONS),
        ("qp_region_max", (C.c_int32 * 3) * GPCC_MAX_QP_REGIONS),
        ("qp_region_offset", (C.c_int32 * 2) * GPCC_MAX_QP_REGIONS),
    ]


class QpRegions(C.Structure):
    """gpcc_qp_regions: the QP regions of a slice for the RAHT slice drivers"""
    _fields_ = [("num_qp_regions", C.c_int32),
                ("qp_region_min", (C.c_int32 * 3) * GPCC_MAX_QP_REGIONS),
                ("qp_region_max", (C.c_int32 * 3) * GPCC_MAX_QP_REGIONS),
                ("qp_region_offset", (C.c_int32 * 2) * GPCC_MAX_QP_REGIONS)]


def qp_regions(regions):
    return set_qp_regions(QpRegions(), regions)


def set_qp_regions(p, regions):
    """regions: [((x0, y0, z0), (x1, y1, z1), (offset_luma, offset_chroma)), ...] -- the boxes' bounds
    inclusive, the first region that contains a point counts (QpSet::regionQpOffset)"""
    assert len(regions) <= GPCC_MAX_QP_REGIONS
    p.num_qp_regions = len(regions)
    for r, (lo, hi, off) in enumerate(regions):
        for k in range(3):
            p.qp_region_min[r][k] = int(lo[k])
            p.qp_region_max[r][k] = int(hi[k])
        p.qp_region_offset[r][0] = int(off[0])
        p.qp_region_offset[r][1] = int(off[1])
    return p


def region_offsets(xyz, regions):
    """the same as a per-point array [n][2] (what gpcc_lift_forward / gpcc_pred_forward take)"""
    import numpy as np
    xyz = np.asarray(xyz)
    out = np.zeros((len(xyz), 2), dtype=np.int32)
    done = np.zeros(len(xyz), dtype=bool)
    for lo, hi, off in regions:
        inside = np.all((xyz >= np.asarray(lo)) & (xyz <= np.asarray(hi)), axis=1) & ~done
        out[inside] = off
        done |= inside
    return out


def lift_params(num_points_in_lod, qp=34, chroma_offset=-1, bitdepth=8, lcp=True, layers=None, scalable=False):
    """Effective values of cfg/octree-liftt-ctc-lossless-geom-lossy-attrs.yaml
    (transformType 2): fixedPointQpOffset = (kFixedPointWeightShift / 2) * 6
    = 24 (quantization.cpp:155-158), last component prediction on."""
    p = LiftParams()
    npl = [int(v) for v in num_points_in_lod]
    assert 1 <= len(npl) <= GPCC_MAX_LODS
    p.num_lods = len(npl)
    for i, v in enumerate(npl):
        p.num_points_in_lod[i] = v
    p.last_component_prediction_enabled_flag = int(lcp)
    p.bitdepth = bitdepth
    lay = layers if layers is not None else [(qp, chroma_offset)]
    p.num_qp_layers = len(lay)
    for i, (a, b) in enumerate(lay):
        p.layer_qp[i][0] = a
        p.layer_qp[i][1] = b
    p.max_qp = 51 + 6 * (bitdepth - 8)
    p.fixed_point_qp_offset = 24
    p.scalable_lifting_enabled_flag = int(scalable)
    return p


class PredParams(C.Structure):
    """ctypes mirror of gpcc_pred_params."""
    _fields_ = [
        ("num_lods", C.c_int32),
        ("num_points_in_lod", C.c_int32 * GPCC_MAX_LODS),
        ("bitdepth", C.c_int32),
        ("num_qp_layers", C.c_int32),
        ("layer_qp", (C.c_int32 * 2) * GPCC_MAX_QP_LAYERS),
        ("max_qp", C.c_int32),
        ("max_num_direct_predictors", C.c_int32),
        ("direct_avg_predictor_disabled_flag", C.c_int32),
        ("adaptive_prediction_threshold", C.c_int32),
        ("inter_component_prediction_enabled_flag", C.c_int32),
        ("quant_neigh_weight", C.c_int32 * 3),
        ("max_num_detail_levels", C.c_int32),
        ("scalable_lifting_enabled_flag", C.c_int32),
        ("num_qp_regions", C.c_int32),
        ("qp_region_min", (C.c_int32 * 3) * GPCC_MAX_QP_REGIONS),
        ("qp_region_max", (C.c_int32 * 3) * GPCC_MAX_QP_REGIONS),
        ("qp_region_offset", (C.c_int32 * 2) * GPCC_MAX_QP_REGIONS),
    ]


def pred_params(num_points_in_lod, qp=34, chroma_offset=0, bitdepth=8, direct=3, avg_disabled=False,
                threshold=64, icp=True, quant_neigh_weight=(0, 0, 0), max_levels=None, layers=None, scalable=False):
    """Effective values of cfg/octree-predt-ctc-lossless-geom-nearlossless-attrs.yaml
    (transformType 1): three direct predictors, adaptivePredictionThreshold 64
    (scaled by the bit depth, hls.h:808-811), inter-component prediction on."""
    p = PredParams()
    npl = [int(v) for v in num_points_in_lod]
    assert 1 <= len(npl) <= GPCC_MAX_LODS
    p.num_lods = len(npl)
    for i, v in enumerate(npl):
        p.num_points_in_lod[i] = v
    p.bitdepth = bitdepth
    lay = layers if layers is not None else [(qp, chroma_offset)]
    p.num_qp_layers = len(lay)
    for i, (a, b) in enumerate(lay):
        p.layer_qp[i][0] = a
        p.layer_qp[i][1] = b
    p.max_qp = 51 + 6 * (bitdepth - 8)
    p.max_num_direct_predictors = direct
    p.direct_avg_predictor_disabled_flag = int(avg_disabled)
    p.adaptive_prediction_threshold = threshold << max(0, bitdepth - 8)
    p.inter_component_prediction_enabled_flag = int(icp)
    for k in range(3):
        p.quant_neigh_weight[k] = quant_neigh_weight[k]
    p.max_num_detail_levels = max_levels if max_levels is not None else max(len(npl), 1)
    p.scalable_lifting_enabled_flag = int(scalable)
    return p


class LodParams(C.Structure):
    """Flattened LoD-generation parameters (AttributeParameterSet LoD fields +
    AttributeBrickHeader::attr_dist2_delta) handed to AttributeLods::generate
    (reference tmc3/AttributeCommon.cpp:44-72)."""
    _fields_ = [
        ("attr_encoding", C.c_int32),            # 1 predicting, 2 lifting
        ("lod_decimation_type", C.c_int32),      # 0 distance, 1 periodic, 2 centroid
        ("num_detail_levels_minus1", C.c_int32),
        ("num_pred_nearest_neighbours_minus1", C.c_int32),
        ("intra_lod_search_range", C.c_int32),
        ("inter_lod_search_range", C.c_int32),
        ("prediction_with_distribution_enabled", C.c_int32),
        ("lod_neigh_bias", C.c_int32 * 3),
        ("intra_lod_prediction_skip_layers", C.c_int32),
        ("dist2", C.c_int32),
        ("attr_dist2_delta", C.c_int32),
        ("canonical_point_order_flag", C.c_int32),
        ("max_points_per_sort_log2_plus1", C.c_int32),
        ("scalable_lifting_enabled_flag", C.c_int32),
        ("max_neigh_range_minus1", C.c_int32),
        ("pred_weight_blending_enabled_flag", C.c_int32),
        ("lod_sampling_period", C.c_int32 * GPCC_MAX_LODS),
    ]


def lod_params(levels=12, decimation=0, dist2=0, dist2_delta=0, neighbours=3, lifting=True,
               distribution=True, bias=(1, 1, 1), inter_range=1100000, intra_range=0,
               sampling_period=4, blend=False):
    """cfg/octree-liftt-ctc-*.yaml after encoder.cpp:799-808 (search ranges
    of -1 become 1100000): 12 detail levels, distance decimation, three
    nearest neighbours, distribution-aware third neighbour."""
    p = LodParams()
    p.attr_encoding = 2 if lifting else 1
    p.lod_decimation_type = decimation
    p.num_detail_levels_minus1 = levels - 1
    p.num_pred_nearest_neighbours_minus1 = neighbours - 1
    p.intra_lod_search_range = intra_range
    p.inter_lod_search_range = inter_range
    p.prediction_with_distribution_enabled = int(distribution)
    for i in range(3):
        p.lod_neigh_bias[i] = bias[i]
    p.intra_lod_prediction_skip_layers = 0x7FFFFFFF
    p.dist2 = dist2
    p.attr_dist2_delta = dist2_delta
    p.pred_weight_blending_enabled_flag = int(blend)
    for i in range(GPCC_MAX_LODS):
        p.lod_sampling_period[i] = sampling_period
    return p


class RecolourParams(C.Structure):
    """ctypes mirror of gpcc_recolour_params."""
    _fields_ = [
        ("dist_offset_fwd", C.c_double), ("dist_offset_bwd", C.c_double),
        ("max_geometry_dist2_fwd", C.c_double), ("max_geometry_dist2_bwd", C.c_double),
        ("max_attribute_dist2_fwd", C.c_double), ("max_attribute_dist2_bwd", C.c_double),
        ("search_range", C.c_int32),
        ("num_neighbours_fwd", C.c_int32), ("num_neighbours_bwd", C.c_int32),
        ("use_dist_weighted_avg_fwd", C.c_int32), ("use_dist_weighted_avg_bwd", C.c_int32),
        ("skip_avg_if_identical_fwd", C.c_int32), ("skip_avg_if_identical_bwd", C.c_int32),
        ("bitdepth", C.c_int32),
    ]


def recolour_params(bitdepth=8, search_range=1, k_fwd=8, k_bwd=1, weighted_fwd=True, weighted_bwd=True,
                    skip_fwd=True, skip_bwd=False, dist_offset_fwd=4.0, dist_offset_bwd=4.0,
                    max_geom_fwd=1000.0, max_geom_bwd=1000.0, max_attr_fwd=1000.0, max_attr_bwd=1000.0):
    """The reference's defaults (TMC3.cpp:1501-1550)."""
    p = RecolourParams()
    p.dist_offset_fwd, p.dist_offset_bwd = dist_offset_fwd, dist_offset_bwd
    p.max_geometry_dist2_fwd, p.max_geometry_dist2_bwd = max_geom_fwd, max_geom_bwd
    p.max_attribute_dist2_fwd, p.max_attribute_dist2_bwd = max_attr_fwd, max_attr_bwd
    p.search_range = search_range
    p.num_neighbours_fwd, p.num_neighbours_bwd = k_fwd, k_bwd
    p.use_dist_weighted_avg_fwd, p.use_dist_weighted_avg_bwd = int(weighted_fwd), int(weighted_bwd)
    p.skip_avg_if_identical_fwd, p.skip_avg_if_identical_bwd = int(skip_fwd), int(skip_bwd)
    p.bitdepth = bitdepth
    return p
