# six fresh processes of tools/adam_mode_probe.py under rocprofv3: the optimiser launch's duration beside the addresses it works on
set -e
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rm -rf gpurun_out/amp_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/amp_$i -- python3 tools/adam_mode_probe.py > gpurun_out/amp_$i.log 2>&1
  python3 - $i <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/amp_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "adam_fused" in r["Name"] or "grad_reduce_all" in r["Name"]:
        print("run", sys.argv[1], r["Name"][23:45], r["Calls"], "avg %.2f us" % (float(r["AverageNs"]) / 1e3))
print("".join(l for l in open("gpurun_out/amp_%s.log" % sys.argv[1]) if l.startswith("PTR") and ("work" in l or "steps" in l or "lr" in l or "scale" in l or "norm" in l)))
PY
done
find gpurun_out -path "*amp_*" -name "*kernel_trace.csv" -delete; find gpurun_out -name "*agent_info.csv" -delete
