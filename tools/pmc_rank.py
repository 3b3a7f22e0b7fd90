#!/usr/bin/env python3
"""Collect SQ / TCC counters for the rank kernels of one config-2-shaped call (GPU box only).
Each counter group runs in its own rocprofv3 --pmc pass (never combined with a trace domain other than the kernel
trace); values are averaged per launch of each kernel and written to gpurun_out/<tag>.json."""
import csv, glob, json, os, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "pmc_rank"
groups = [g.split(",") for g in (sys.argv[2].split(";") if len(sys.argv) > 2 else [
    "SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY",
    "SQ_INSTS_VALU,SQ_INSTS_LDS,SQ_INSTS_SALU,SQ_INSTS_VMEM",
    "SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_LDS",
    "SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_BRANCH,SQ_ACTIVE_INST_SCA",
])]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for gi, g in enumerate(groups):
    d = os.path.join(root, "gpurun_out", f"{tag}_g{gi}")
    target = os.environ.get("PMC_CMD", "tools/time_rank.py 10000 1000 3").split()
    cmd = ["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(root, target[0])] + target[1:]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=170)
    print(f"group {gi} {g}: rc={r.returncode}", flush=True)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "sd::" not in name:
                continue
            kn = name.split("(")[0].replace("void ", "")
            key = (kn, row["Counter_Name"])
            a = acc.setdefault(key, [0.0, set()])
            a[0] += float(row["Counter_Value"])
            a[1].add(row["Dispatch_Id"])
        for (kn, cn), (v, ds) in acc.items():
            out.setdefault(kn, {})[cn] = v / max(1, len(ds))
json.dump(out, open(os.path.join(root, "gpurun_out", f"{tag}.json"), "w"), indent=1)
for kn, cs in out.items():
    print(kn)
    for cn, v in sorted(cs.items()):
        print(f"   {cn:28s} {v:16.0f}")
