#!/bin/bash
# VGPRs / spills / scratch / LDS of every kernel of one translation unit, from the code object's notes (no GPU needed).
# usage: bash tools/kernel_resources.sh bez_isaacgym_amd/csrc/bez_step_ws8.hip
set -e
SRC=${1:-bez_isaacgym_amd/csrc/bez_step_ws8.hip}
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --cuda-device-only -c -o $T/k.co "$SRC"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.elf | grep -E "\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|group_segment_fixed" | paste - - - - - | sed 's/ \+/ /g'
rm -rf $T
