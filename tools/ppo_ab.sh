#!/bin/bash
# A/B of the PPO leg between library builds (tools/ab_build.sh) on the SAME box: devices differ by several percent between gpurun calls.
# usage: bash tools/ppo_ab.sh a b ...     (build_ab/a.so, build_ab/b.so; the in-tree library is swapped per run and restored at the end)
set -e
L=bez_isaacgym_amd/lib/libbez_sim.so
cp $L /tmp/libbez_sim.keep
for round in 1 2 3; do
  for n in "$@"; do
    cp build_ab/$n.so $L
    python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 --ppo-epochs 30 ${PPO_AB_FLAGS} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', round(d['ppo']['value']), '%.3f ms' % d['ppo']['epoch_ms'])"
  done
done
cp /tmp/libbez_sim.keep $L
