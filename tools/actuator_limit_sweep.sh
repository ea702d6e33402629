#!/bin/bash
# Does the trained reward depend on the actuator limits (PhysX may not enforce dof 'velocity' = 2 pi on the GPU pipeline; the reference's own
# comment next to it says 24.5 rad/s)?  1500-epoch trainings, mean reward of the last 20 logged epochs.  usage (GPU box): bash tools/actuator_limit_sweep.sh [outfile]
OUT=${1:-gpurun_out/r03_actuator_limits.txt}
: > $OUT
B=task.sim.bez
run() {
  name=$1; shift
  r=$(timeout -k 10 400 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=1500 "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {if (n) printf "%.2f", s/n; else printf "nan"}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
for seed in 42 43; do
run vel24_s$seed seed=$seed +$B.vel_limit=24.5
run effort5_s$seed seed=$seed +$B.effort=5
run vel24_effort5_s$seed seed=$seed +$B.vel_limit=24.5 +$B.effort=5
run novel_s$seed seed=$seed +$B.vel_limit=1000
done
