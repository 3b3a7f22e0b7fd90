#!/usr/bin/env python3
"""SQ counters of the strict-depth kernels (GPU box): VALU wave-instructions, issue share, waits, per launch.
usage: issue_strict.py <tag> [n] [T]   (writes gpurun_out/<tag>_issue_strict.json)"""
import csv, glob, json, os, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
n = sys.argv[2] if len(sys.argv) > 2 else "10000"
T = sys.argv[3] if len(sys.argv) > 3 else "1000"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, TMPDIR="/tmp")
d = os.path.join(root, "gpurun_out", f"{tag}_iss_strict")
cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
       "SQ_INSTS_SALU", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
       os.path.join(root, "tools", "time_strict.py"), n, T, "walks"]
r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
print("rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], flush=True)
acc, durs = {}, {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "strict" not in k:
            continue
        a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, set()])
        a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "strict" in k:
            durs.setdefault(k, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
out = {"workload": f"tools/time_strict.py {n} {T} walks", "note": "per launch; SQ_* as rocprofv3 reports them"}
for k, cs in acc.items():
    c = {name: v[0] / max(1, len(v[1])) for name, v in cs.items()}
    us = sum(durs.get(k, [0])) / max(1, len(durs.get(k, [0])))
    out[k] = {"kernel_us_under_profiler": us, "launches": len(durs.get(k, [])), "counters": c,
              "valu_issue_share": c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1)),
              "wait_share": c.get("SQ_WAIT_INST_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1)),
              "valu_wave_instr_per_us": c.get("SQ_INSTS_VALU", 0) / max(us, 1e-9)}
    print(k, json.dumps(out[k]))
json.dump(out, open(os.path.join(root, "gpurun_out", f"{tag}_issue_strict.json"), "w"), indent=1)
