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
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)  # gpurun merges into the old scratch tree: newest wins
    return hits[-1] if hits else None


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


def kernel_code_bytes(kernel_mangled, unit="bez_step_ws8.hip"):
    """codeLenInByte of one kernel: device-only assembly of its translation unit with the build's own flags."""
    import tempfile
    from bez_isaacgym_amd.build import _flags
    src = os.path.join(ROOT, "bez_isaacgym_amd", "csrc", unit)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "ws.s")
        flags = [f for f in _flags() if f != "-fPIC"]
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", out, src], capture_output=True, text=True)
        if r.returncode != 0:
            return None
        name = None
        for line in open(out):
            if line.startswith("\t.globl\t"):
                name = line.split()[1]
            elif line.startswith("; codeLenInByte") and name == kernel_mangled:
                return int(line.split("=")[1])
    return None


def size_sweep(directory, counter):
    """mean counter value (KiB) per launch of the step kernel, by number of envs (grid size / 512 threads x the kernel's envs per workgroup)."""
    f = _one(os.path.join(directory, "**", "*counter_collection.csv"))
    if not f:
        return None
    acc = {}
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == counter and KERNEL in row["Kernel_Name"]:
            n = int(row["Grid_Size"]) // 512 * (16 if "w8q" in row["Kernel_Name"] else 64)   # envs per workgroup: 64 (ws8) / 16 (ws8q, four lanes per env)
            acc.setdefault(n, {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[n][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {n: (lambda v: sum(v) / len(v))(list(d.values())[4:]) for n, d in sorted(acc.items())}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
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
    fp = _one(os.path.join(src, tag + "_ppo_stats", "**", "*kernel_stats.csv"))
    if fp:
        rows = list(csv.reader(open(fp)))
        with open(os.path.join(out_dir, tag + "_ppo_kernel_stats.csv"), "w", newline="") as g:
            w = csv.writer(g)
            for r in rows[:40]:
                r[0] = r[0][:110]
                w.writerow(r)
    fp = _one(os.path.join(src, tag + "_ppo_dp_stats", "**", "*kernel_stats.csv"))
    if fp:   # the PPO leg twice: single-GPU path, then the data-parallel path (bench.py --dp-path); the kernels only the latter runs show up here
        rows = list(csv.reader(open(fp)))
        with open(os.path.join(out_dir, tag + "_ppo_dp_kernel_stats.csv"), "w", newline="") as g:
            w = csv.writer(g)
            for r in rows[:40]:
                r[0] = r[0][:110]
                w.writerow(r)
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
    # 3. attribution: FETCH_SIZE(N) = (XCDs touched) x fixed + N x per_env.  The fixed part is the kernel's instruction stream,
    # fetched once into each XCD's L2 per launch and tallied 1:1 (64-B requests); only the per-env part (4 B per lane, 256-B wave
    # rows = 128-B requests) takes the x2 correction.
    sw_r = size_sweep(os.path.join(src, tag + "_sweep_fetch"), "FETCH_SIZE")
    sw_w = size_sweep(os.path.join(src, tag + "_sweep_write"), "WRITE_SIZE")
    if sw_r and 4096 in sw_r and 16384 in sw_r:
        per_env_raw = (sw_r[16384] - sw_r[4096]) * 1024 / (16384 - 4096)
        fixed_raw = (sw_r[4096] * 1024 - 4096 * per_env_raw) / 8
        mangled = kernel_name.split("(")[0] if kernel_name else None
        quad = bool(kernel_name) and "w8q" in kernel_name
        code = (kernel_code_bytes("_ZN3bez3w8q15step_kernel_ws8ILb1ELb1ELb0ELb0EEEvNS_6ParamsE", "bez_step_ws8q.hip") if quad
                else kernel_code_bytes("_ZN3bez2w815step_kernel_ws8ILb1ELb1ELb0ELb0EEEvNS_6ParamsE"))
        att = {"FETCH_SIZE_KiB_by_num_envs": sw_r, "WRITE_SIZE_KiB_by_num_envs": sw_w,
               "fit": "raw FETCH_SIZE bytes = 8 XCDs x fixed + num_envs x per_env (from the 4096 and 16384 points)",
               "fixed_bytes_per_xcd_raw": fixed_raw, "kernel_codeLenInByte": code,
               "one_workgroup_launch_raw_bytes": sw_r.get(64) and sw_r[64] * 1024,
               "per_env_read_bytes_raw": per_env_raw, "per_env_read_bytes_corrected": per_env_raw * rf,
               "per_env_write_bytes": sw_w and sw_w[4096] * 1024 * wf / 4096,
               "kernel_stores_per_env": {"state 62 f32": 248, "dof targets": 72, "obs 54": 216, "reward": 4, "reset/progress/timeout": 24, "sum": 564,
                                         "skipped under BEZ_FLAG_LEAN_STEP (bench.py default)": {"prev_lin_vel": 12, "net contact force 22x3": 264, "feet flags 8": 32}},
               "algorithmic_read_per_env": 336, "algorithmic_write_per_env": 492}
        res["attribution"] = att
        if fetch and write and code and abs(fixed_raw - code) < 0.05 * code:
            data_raw = fetch["mean"] * 1024 - 8 * fixed_raw
            rd = 8 * fixed_raw + data_raw * rf
            res["hbm_bytes_per_launch_naive_x2"] = dict(res["hbm_bytes_per_launch"])
            res["hbm_bytes_per_launch"] = {"read": rd, "read_instruction_stream": 8 * fixed_raw, "read_data": data_raw * rf, "write": wr, "total": rd + wr,
                                           "note": "FETCH_SIZE split by the size sweep: instruction stream (%.0f B x 8 XCD L2s, = codeLenInByte %d within 5 %%, "
                                                   "tallied 1:1) + per-env data x %.3f; WRITE_SIZE x %.3f" % (fixed_raw, code, rf, wf)}
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
