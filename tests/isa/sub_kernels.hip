// tests/isa/sub_kernels.hip -- TEST INFRASTRUCTURE: the sub-node level kernels of the product
// (mpeg-pcc-tmc13_amd/csrc/raht_subnode.hpp) instantiated on their own, with the product's flags, so that
// tests/test_isa_chain_discipline.py can read their gfx950 ISA without compiling the whole library
// (minutes).  Same template, same arguments, same compiler: the same machine code.
#include <hip/hip_runtime.h>

#include "raht_subnode.hpp"

namespace gpcc {
#define GPCC_ISA_INSTANCE(C, MODE, A) template __global__ void raht_level_sub_kernel<C, MODE, A, false>(LevelCtx)
GPCC_ISA_INSTANCE(1, kSynth, ArithI64);
GPCC_ISA_INSTANCE(1, kSynth, ArithF64);
GPCC_ISA_INSTANCE(1, kFused, ArithI64);
GPCC_ISA_INSTANCE(1, kLossySub, ArithI64);
GPCC_ISA_INSTANCE(1, kLossySub, ArithF64);
GPCC_ISA_INSTANCE(3, kSynth, ArithI64);
GPCC_ISA_INSTANCE(3, kSynth, ArithF64);
GPCC_ISA_INSTANCE(3, kLossySub, ArithI64);
GPCC_ISA_INSTANCE(3, kLossySub, ArithF64);
#define GPCC_ISA_REC_INSTANCE(C, MODE, A) template __global__ void raht_level_sub_kernel<C, MODE, A, false, true>(LevelCtx)
GPCC_ISA_REC_INSTANCE(1, kSynth, ArithF64);
GPCC_ISA_REC_INSTANCE(1, kLossySub, ArithF64);
GPCC_ISA_REC_INSTANCE(1, kLossySub, ArithI64);
GPCC_ISA_REC_INSTANCE(3, kSynth, ArithI64);
}  // namespace gpcc
