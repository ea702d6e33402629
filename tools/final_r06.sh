#!/bin/bash
# Round 6's measurement pass on the GPU box, default kernel (run AFTER profiles/r06_pmc_traffic.json of the same build is committed):
# bench lines, same-box A/B against round 5's library, and the learning / sim-to-sim artefacts.   bash tools/final_r06.sh
O=gpurun_out/r06f; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_form.json 2> /dev/null
python bench.py --randomize --no-cpu-baseline --no-dp-path > $O/bench_randomize.json 2> /dev/null
{ python tools/ab_bench.py build_ab/r05.so build_ab/r06.so build_ab/r06.so:ws8
  echo "bench.py lean step, same box: round-5 library (BEZ_SIM_LIB), this build's one-lane form (BEZ_SIM_KERNEL=ws8), this build's default: value ms_per_step kernel_ms"
  for v in "BEZ_SIM_LIB=build_ab/r05.so" "BEZ_SIM_KERNEL=ws8" "X=1"; do
    env $v python bench.py --no-cpu-baseline --ppo-epochs 0 --no-full-store 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done; } > $O/ab_same_box.txt 2>&1
[ "$1" = "bench" ] && { echo done; exit 0; }
python tools/seed_table.py --seeds 42 43 44 45 46 47 48 49 --epochs 6156 --out $O/seed_table.txt > $O/seed_table.log 2>&1
python tools/sim2sim_gpu.py --envs 256 --out $O/sim2sim_256.json > $O/sim2sim_256.log 2>&1
python tools/sim2sim_gpu.py --out $O/sim2sim.json > $O/sim2sim.log 2>&1
echo done
