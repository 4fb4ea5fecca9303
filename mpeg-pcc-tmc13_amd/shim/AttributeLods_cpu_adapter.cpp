// AttributeLods_cpu_adapter.cpp -- compiled with -Dgenerate=generateCpu, like
// the reference's tmc3/AttributeCommon.cpp (see AttributeLods_mi355.cpp and
// INTEGRATION.md): in this translation unit `lods.generate(...)` therefore
// names the reference's own, renamed, CPU implementation.
#include "AttributeCommon.h"

namespace gpcc_shim {
void
lods_generate_cpu(
  pcc::AttributeLods& lods, const pcc::AttributeParameterSet& aps,
  const pcc::AttributeBrickHeader& abh, int geom_num_points_minus1,
  int minGeomNodeSizeLog2, const pcc::PCCPointSet3& cloud,
  const pcc::AttributeInterPredParams& attrInterPredParams)
{
  lods.generate(
    aps, abh, geom_num_points_minus1, minGeomNodeSizeLog2, cloud,
    attrInterPredParams);
}
}  // namespace gpcc_shim
