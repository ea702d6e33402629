set -e
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_ppo_fused.py -m gpu -x -q > gpurun_out/ab32_tests.log 2>&1 || { tail -30 gpurun_out/ab32_tests.log; exit 1; }
tail -3 gpurun_out/ab32_tests.log
for v in 64 32 64 32; do
  export BEZ_PF_ROWS=$v
  rm -rf gpurun_out/kab_r$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kab_r$v -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dp-path --ppo-epochs 10 > gpurun_out/kab_r$v.log 2>&1
  echo "== rows $v"
  python3 - "$v" <<'PY'
import csv, glob, sys, json
f = glob.glob("gpurun_out/kab_r%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "policy_forward" in r["Name"]:
        print("%-100s calls %6s avg %9.2f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
l = [x for x in open("gpurun_out/kab_r%s.log" % sys.argv[1]) if x.startswith("{")][-1]
j = json.loads(l); print("ppo", j.get("ppo", {}).get("value"), j.get("ppo", {}).get("ms_per_epoch"))
PY
done
find gpurun_out -name "*agent_info.csv" -delete; find gpurun_out -name "*kernel_trace.csv" -path "*kab_*" -delete
