#!/bin/bash
# A/B of the PPO leg between library builds (tools/ab_build.sh) on the SAME box: devices differ by several percent between gpurun calls.
# usage: bash tools/ppo_ab.sh a b ...     (build_ab/a.so, build_ab/b.so, selected through BEZ_SIM_LIB: the in-tree library is never touched)
set -e
for round in 1 2 3; do
  for n in "$@"; do
    BEZ_SIM_LIB=$PWD/build_ab/$n.so python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 --ppo-epochs 30 ${PPO_AB_FLAGS} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', round(d['ppo']['value']), '%.3f ms' % d['ppo']['epoch_ms'])"
  done
done
