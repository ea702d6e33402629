#!/bin/bash
# Runs on the GPU box (gpurun): bench line, rocprofv3 kernel stats, and the HBM / SQ counters in their own passes.
# usage: bash tools/collect_profiles.sh r03      (raw output under gpurun_out/<tag>_*; summarise with tools/summarize_profiles.py)
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
O=gpurun_out
B="python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --ppo-epochs 0 --no-full-store"
S="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --ppo-epochs 0 --no-full-store"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $B > $O/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $S > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $S > $O/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/${TAG}_pmc_sq1 -- $S > $O/${TAG}_pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/${TAG}_pmc_sq2 -- $S > $O/${TAG}_pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_cal_fetch -- python3 tools/pmc_calibrate.py > $O/${TAG}_cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_cal_write -- python3 tools/pmc_calibrate.py > $O/${TAG}_cal_write.log 2>&1
# size sweep (1 / 8 / 64 / 256 workgroups): separates the per-launch fixed part of FETCH_SIZE (the instruction stream, once per XCD L2) from the per-env part
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_sweep_fetch -- python3 tools/pmc_probe.py > $O/${TAG}_sweep_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_sweep_write -- python3 tools/pmc_probe.py > $O/${TAG}_sweep_write.log 2>&1
# the PPO leg (BASELINE.json configs[2]): kernel stats of 10 + 4 epochs
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ppo_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dp-path > $O/${TAG}_ppo_stats.log 2>&1
# the same leg on the data-parallel code path (1-rank RCCL group, BEZ_PPO_FORCE_DIST=1): which kernels a rank of an N-GPU job runs per epoch
# (r05's committed table was taken with both legs in one process; bench.py has since moved the extra leg into a child process, profiled here directly)
MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ppo_dp_stats -- python3 bench.py --dp-path-child --ppo-epochs 30 > $O/${TAG}_ppo_dp_stats.log 2>&1
# keep only the CSVs the summary needs (the merge back is capped at 64 MiB)
find $O -name "*agent_info.csv" -delete
echo collected
