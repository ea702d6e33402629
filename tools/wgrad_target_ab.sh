set -e
export TMPDIR=/tmp
for v in "250 32" "254 40" "250 32" "254 40" "248 32"; do
  set -- $v
  export BEZ_WGRAD_TARGET_WGS=$1 BEZ_WGRAD_NSPLIT=$2
  rm -rf gpurun_out/kab_w
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kab_w -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dp-path --ppo-epochs 10 > gpurun_out/kab_w.log 2>&1
  python3 - "$v" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/kab_w/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "wgrad_kernel" in r["Name"] or "grad_reduce_all" in r["Name"]:
        print(sys.argv[1], r["Name"][23:40], r["Calls"], "avg %.2f us" % (float(r["AverageNs"]) / 1e3))
PY
done
find gpurun_out/kab_w -name "*.csv" -delete
