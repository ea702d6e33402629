#!/bin/bash
# usage (GPU box): bash tools/drive_gain_sweep.sh [outfile]  -- 1500-epoch trainings over the PD gains / armature at the nominal effort and speed limits
OUT=${1:-gpurun_out/r03_drive_gains.txt}
: > $OUT
run() {
  name=$1; shift
  r=$(timeout -k 10 400 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=1500 "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {if (n) printf "%.2f", s/n; else printf "nan"}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
for seed in 42 43; do
run kd2_s$seed seed=$seed task.env.control.damping=2
run kd4_s$seed seed=$seed task.env.control.damping=4
run kd15_s$seed seed=$seed task.env.control.damping=15
run kp200_s$seed seed=$seed task.env.control.stiffness=200
run kp50_kd4_s$seed seed=$seed task.env.control.stiffness=50 task.env.control.damping=4
run jfric0_s$seed seed=$seed +task.sim.bez.joint_friction=0
done
