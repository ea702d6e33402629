#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/r02_*) into the committed summaries:
  profiles/<tag>_rocprofv3_kernel_stats.csv   the --kernel-trace --stats table (long torch kernel names truncated)
  profiles/<tag>_pmc_traffic.json             HBM bytes per launch of the dominant kernel from the separate FETCH_SIZE /
                                              WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2
                                              on gfx950; the factor is re-measured by the 1 GiB calibration kernels), plus SQ counters
The JSON records the kernel name and the git revision + source hash of the library it was measured on, so bench.py only
reports `roofline.traffic` when the profile belongs to the build that is running."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNEL = "step_kernel_ws"


def _one(pattern):
    hits = glob.glob(pattern, recursive=True)
    return hits[0] if hits else None


def counter_mean(directory, counter, kernel_substr):
    f = _one(os.path.join(directory, "**", "*counter_collection.csv"))
    if not f:
        return None
    vals = {}
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
            vals.setdefault(row["Dispatch_Id"], 0.0)
            vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
    if not vals:
        return None
    v = list(vals.values())
    return {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "dispatches": len(v)}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out")
    out_dir = os.path.join(ROOT, "profiles")
    from bez_isaacgym_amd.build import source_hash
    rev = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    # 1. kernel stats
    f = _one(os.path.join(src, tag + "_stats", "**", "*kernel_stats.csv"))
    kernel_name, avg_ns = None, None
    if f:
        rows = list(csv.reader(open(f)))
        with open(os.path.join(out_dir, tag + "_rocprofv3_kernel_stats.csv"), "w", newline="") as g:
            w = csv.writer(g)
            for r in rows[:14]:
                r[0] = r[0][:110]
                w.writerow(r)
        for r in rows[1:]:
            if KERNEL in r[0]:
                kernel_name, avg_ns = r[0], float(r[3])
                break
    # 2. PMC
    res = {"kernel": kernel_name, "kernel_avg_ns_rocprofv3": avg_ns, "num_envs": 4096, "git_rev": rev, "source_hash": source_hash(),
           "commands": open(os.path.join(ROOT, "tools", "collect_profiles.sh")).read().splitlines()}
    fetch = counter_mean(os.path.join(src, tag + "_pmc_fetch"), "FETCH_SIZE", KERNEL)
    write = counter_mean(os.path.join(src, tag + "_pmc_write"), "WRITE_SIZE", KERNEL)
    cal_r = counter_mean(os.path.join(src, tag + "_cal_fetch"), "FETCH_SIZE", "calib_read")
    cal_w = counter_mean(os.path.join(src, tag + "_cal_write"), "WRITE_SIZE", "calib_write")
    true_kib = 256 * 1024 * 1024 * 4 / 1024.0
    res["raw_KiB"] = {"FETCH_SIZE": fetch, "WRITE_SIZE": write}
    rf = true_kib / cal_r["mean"] if cal_r else 2.0
    wf = true_kib / cal_w["mean"] if cal_w else 1.0
    res["calibration"] = {"read_1GiB_dword_per_lane": {"FETCH_SIZE_KiB": cal_r and cal_r["mean"], "true_KiB": true_kib, "factor": rf},
                          "write_1GiB_dword_per_lane": {"WRITE_SIZE_KiB": cal_w and cal_w["mean"], "true_KiB": true_kib, "factor": wf}}
    if fetch and write:
        rd, wr = fetch["mean"] * 1024 * rf, write["mean"] * 1024 * wf
        res["hbm_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr,
                                       "note": "FETCH_SIZE x %.3f, WRITE_SIZE x %.3f (factors measured by the calibration kernels of the same run; "
                                               "MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at 64 B)" % (rf, wf)}
    res["algorithmic_bytes_per_launch"] = 828 * 4096
    sq = {}
    for d in glob.glob(os.path.join(src, tag + "_pmc_sq*")):
        f = _one(os.path.join(d, "**", "*counter_collection.csv"))
        if not f:
            continue
        acc = {}
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for k, v in acc.items():
            sq[k] = sum(v.values()) / len(v)
    res["sq_counters_per_launch"] = sq
    with open(os.path.join(out_dir, tag + "_pmc_traffic.json"), "w") as g:
        json.dump(res, g, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "commands"}, indent=1))


if __name__ == "__main__":
    main()
