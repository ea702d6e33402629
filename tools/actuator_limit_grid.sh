#!/bin/bash
# usage (GPU box): bash tools/actuator_limit_grid.sh [outfile]   -- 1500-epoch trainings over (vel_limit, effort), 2 seeds
OUT=${1:-gpurun_out/r03_actuator_grid.txt}
: > $OUT
B=task.sim.bez
run() {
  name=$1; shift
  r=$(timeout -k 10 400 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=1500 "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {if (n) printf "%.2f", s/n; else printf "nan"}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
for seed in 42 43; do
for v in 12 24.5; do
for e in 3.5 5 8; do
run vel${v}_effort${e}_s$seed seed=$seed +$B.vel_limit=$v +$B.effort=$e
done; done
run vel1000_effort5_s$seed seed=$seed +$B.vel_limit=1000 +$B.effort=5
run vel1000_effort100_s$seed seed=$seed +$B.vel_limit=1000 +$B.effort=100
done
