#!/bin/bash
# asm_patch_build.sh <patched device .s> <out.so>: re-assembles the compiler's own device assembly (hipcc --cuda-device-only -S) after an edit -- code object -> fat binary -> host object -> library (run from exp/asm; used by the finish-kernel experiment, profiles/r04_finish_lds_root_cause.txt)
set -e
s=$1; out=$2; b=${s%.s}
L=/opt/rocm/lib/llvm/bin
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $s -o $b.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $b.out $b.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$b.out -output=$b.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I../../include -I../../mpeg-pcc-tmc13_amd/csrc -DGPCC_FIN_VAR=1 --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $b.hipfb -c ../../mpeg-pcc-tmc13_amd/csrc/gpcc_attr_mi355.hip -o $b.host.o
/opt/rocm/bin/hipcc -shared -fPIC $b.host.o -o $out
rm -f $b.o $b.out $b.hipfb $b.host.o
