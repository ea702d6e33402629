import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.8):]
idx = [i for i, r in enumerate(rows) if "step_kernel_ws8" in r["Kernel_Name"]]
# first pair of consecutive sim kernels that are close (within the rollout)
for a, b in zip(idx, idx[1:]):
    if b - a < 12:
        for r in rows[a:b + 1]:
            print("%8.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:110]))
        break
