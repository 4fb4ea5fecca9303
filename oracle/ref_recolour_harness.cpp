// ref_recolour_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// extern "C" marshalling around the reference's own pcc::recolour
// (tmc3/pointset_processing.cpp:926-957), compiled with the unmodified reference
// sources into oracle/_ref/libtmc3_ref.so (oracle/Makefile).  No reference source
// text is copied here.
#include <cstdint>

#include "PCCPointSet.h"
#include "hls.h"
#include "pointset_processing.h"

#include "gpcc_attr_mi355.h"

extern "C" int
ref_recolour(
  const gpcc_recolour_params* p, const int32_t* src_xyz, const int32_t* src_attrs, int32_t ns,
  const int32_t* tgt_xyz, int32_t nt, int32_t c, float scale, const int32_t offset[3],
  int32_t* tgt_attrs)
{
  pcc::PCCPointSet3 source, target;
  source.resize(ns);
  target.resize(nt);
  if (c == 3)
    source.addColors();
  else
    source.addReflectances();
  for (int i = 0; i < ns; i++) {
    source[i] = pcc::Vec3<int32_t>(src_xyz[3 * i], src_xyz[3 * i + 1], src_xyz[3 * i + 2]);
    if (c == 3)
      source.setColor(
        i, pcc::Vec3<pcc::attr_t>(
             (pcc::attr_t)src_attrs[3 * i], (pcc::attr_t)src_attrs[3 * i + 1],
             (pcc::attr_t)src_attrs[3 * i + 2]));
    else
      source.setReflectance(i, (pcc::attr_t)src_attrs[i]);
  }
  for (int i = 0; i < nt; i++)
    target[i] = pcc::Vec3<int32_t>(tgt_xyz[3 * i], tgt_xyz[3 * i + 1], tgt_xyz[3 * i + 2]);

  pcc::AttributeDescription desc{};
  desc.attr_num_dimensions_minus1 = c - 1;
  desc.bitdepth = p->bitdepth;
  desc.attributeLabel =
    c == 3 ? pcc::KnownAttributeLabel::kColour : pcc::KnownAttributeLabel::kReflectance;
  pcc::RecolourParams cfg;
  cfg.distOffsetFwd = p->dist_offset_fwd;
  cfg.distOffsetBwd = p->dist_offset_bwd;
  cfg.maxGeometryDist2Fwd = p->max_geometry_dist2_fwd;
  cfg.maxGeometryDist2Bwd = p->max_geometry_dist2_bwd;
  cfg.maxAttributeDist2Fwd = p->max_attribute_dist2_fwd;
  cfg.maxAttributeDist2Bwd = p->max_attribute_dist2_bwd;
  cfg.searchRange = p->search_range;
  cfg.numNeighboursFwd = p->num_neighbours_fwd;
  cfg.numNeighboursBwd = p->num_neighbours_bwd;
  cfg.useDistWeightedAvgFwd = p->use_dist_weighted_avg_fwd != 0;
  cfg.useDistWeightedAvgBwd = p->use_dist_weighted_avg_bwd != 0;
  cfg.skipAvgIfIdenticalSourcePointPresentFwd = p->skip_avg_if_identical_fwd != 0;
  cfg.skipAvgIfIdenticalSourcePointPresentBwd = p->skip_avg_if_identical_bwd != 0;
  const pcc::point_t off(offset[0], offset[1], offset[2]);
  const int rc = pcc::recolour(desc, cfg, source, scale, off, &target);
  if (rc)
    return rc;
  for (int i = 0; i < nt; i++) {
    if (c == 3) {
      const auto col = target.getColor(i);
      for (int k = 0; k < 3; k++)
        tgt_attrs[3 * i + k] = col[k];
    } else {
      tgt_attrs[i] = target.getReflectance(i);
    }
  }
  return 0;
}

// The sort pcc::recolour applies to its backward lists (pointset_processing.cpp:416-422:
// std::sort with a comparator on the distance alone) on a caller's (distance, source) pairs --
// what the restated sort of oracle/recolour_oracle.c is compared with.
#include <algorithm>
#include <vector>

extern "C" void
ref_std_sort_by_dist(double* dist, int32_t* src, int32_t n)
{
  struct DistSrc {
    double dist;
    int32_t src;
  };
  std::vector<DistSrc> v(n);
  for (int i = 0; i < n; i++)
    v[i] = DistSrc{dist[i], src[i]};
  std::sort(v.begin(), v.end(), [](const DistSrc& a, const DistSrc& b) { return a.dist < b.dist; });
  for (int i = 0; i < n; i++) {
    dist[i] = v[i].dist;
    src[i] = v[i].src;
  }
}
