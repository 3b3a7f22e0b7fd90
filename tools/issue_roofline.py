#!/usr/bin/env python3
"""Issue-based roofline of the VALU-bound secondary kernels (GPU box): SQ counters per launch (rocprofv3 --pmc, kernel trace
only) -> VALU wave-instructions per unit of work and the share of the SIMDs' issue cycles they take.
usage: issue_roofline.py <tag>   (writes gpurun_out/<tag>_issue_roofline.json)"""
import csv, glob, json, os, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, TMPDIR="/tmp")
UNITS = {"simplex8": (500 * 256 * 50, "sxw_apply_kernel<8>"), "simplex3": (100000 * 256, "sxw_apply_kernel<3>"),
         "l1": (1e10, "l1_depth_kernel<3>"), "strict": (2000 * 1999 * 1998 / 2, "strict_pairs2_kernel")}
out = {"note": "per launch of the named kernel; SQ_* in wave-instructions / quad-cycles as rocprofv3 reports them (gfx950: SQ_WAVE_CYCLES, "
               "SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU count quad-cycles summed over SIMDs).  valu_per_unit = SQ_INSTS_VALU * 64 lanes / units "
               "is the lane-instruction count per unit of work; valu_issue_share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of the "
               "waves' cycles spent issuing VALU)."}
for what, (units, kname) in UNITS.items():
    d = os.path.join(root, "gpurun_out", f"{tag}_iss_{what}")
    cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
           "SQ_INSTS_SALU", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(root, "tools", "time_secondary.py"), what, "2"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    print(what, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], flush=True)
    acc, durs = {}, []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kname.split("<")[0] not in row["Kernel_Name"]:
                continue
            a = acc.setdefault(row["Counter_Name"], [0.0, set()])
            a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kname.split("<")[0] in row["Kernel_Name"]:
                durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    c = {k: v[0] / max(1, len(v[1])) for k, v in acc.items()}
    if not c:
        continue
    us = sum(durs) / max(1, len(durs))
    out[what] = {"kernel": kname, "units_per_launch": units, "kernel_us_under_profiler": us, "counters": c,
                 "valu_lane_instructions_per_unit": c.get("SQ_INSTS_VALU", 0) * 64 / units,
                 "valu_issue_share": c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1)),
                 "units_per_s": units / (us * 1e-6)}
    print("  ", out[what]["valu_lane_instructions_per_unit"], out[what]["valu_issue_share"], us)
json.dump(out, open(os.path.join(root, "gpurun_out", f"{tag}_issue_roofline.json"), "w"), indent=1)
