#!/bin/bash
# VGPRs / spills of each role of the wave-specialised step kernel compiled ALONE (offline, no GPU): which role sits at the register ceiling.
# usage: [ROLE_TU=bez_step_ws8q] bash tools/role_resources.sh [extra hipcc flags]
for r in 0 1 2 3 4 5 6 7; do
  ( T=$(mktemp -d)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --cuda-device-only -DBEZ_AB_ONLY_ROLE=$r "$@" -c -o $T/k.co bez_isaacgym_amd/csrc/${ROLE_TU:-bez_step_ws8}.hip 2>/dev/null
    /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.elf
    echo "role $r: $(/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.elf | grep -E "\.name:|\.vgpr_count|vgpr_spill" | paste - - - | grep "ILb1ELb1ELb0ELb0" | sed 's/.*vgpr_count: *\([0-9]*\).*vgpr_spill_count: *\([0-9]*\).*/vgpr \1 spills \2/')"
    rm -rf $T ) &
done; wait
