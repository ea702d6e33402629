#!/bin/bash
# same-box A/B of bez_wgrad.hip variants (tools/ab_build.sh WORK <name> -DBEZ_WGRAD_TPW=8 ...): standalone kernel + reduction times and the exact tests
# usage: bash tools/wgrad_ab.sh name[:nsplit] ...
for spec in "$@"; do
  v=${spec%%:*}; ns=${spec#*:}; [ "$ns" = "$spec" ] && ns=32
  echo "== $v nsplit $ns"
  BEZ_WGRAD_NSPLIT=$ns BEZ_SIM_LIB=$PWD/build_ab/$v.so python tools/wgrad_bench.py all L1 2>&1 | grep -v amdgpu
  BEZ_WGRAD_NSPLIT=$ns BEZ_SIM_LIB=$PWD/build_ab/$v.so python -m pytest tests/test_gpu_ppo_fused.py -m gpu -q -k "wgrad" 2>&1 | tail -1
done
