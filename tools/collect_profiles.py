#!/usr/bin/env python3
"""Collect the rocprofv3 evidence a round's numbers rest on (GPU box only; writes gpurun_out/<tag>_*):

  kernel stats   rocprofv3 --kernel-trace --stats of bench.py (headline, rotating matrices), of config 3 on one GPU, of the
                 strict path (10 000 x 1 000; 10^6 points in R^3) and of the medium route (20 000 x 500)
  HBM traffic    rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes (kernel trace only, no other trace domain),
                 corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x 2 on gfx950, calibrated in the same session on
                 sd::nan_count_rows_kernel (bench.py --algo pairwise), which reads the 80 000 000-byte matrix exactly once.

usage: collect_profiles.py <tag> [workload ...]       e.g. r03 (everything) or r03b config3 (stats + traffic of one)
"""
import shutil
import csv
import glob
import json
import os
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
env = dict(os.environ, TMPDIR="/tmp")
PY = sys.executable

WORKLOADS = {
    "bench": ([os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "4", "--no-cpu-baseline", "--no-extras"],
              "10000x1000", "bench.py headline: config 2, 4 matrices in rotation (every step streams from HBM)"),
    "config3": ([os.path.join(root, "tools", "time_rank.py"), "100000", "256", "10"], "100000x256",
                "config 3 on one GPU: 100000 curves x 256 timepoints, large-n route"),
    "strict": ([os.path.join(root, "tools", "time_strict.py"), "10000", "1000", "walks"], "10000x1000",
               "strict band depth (relax=False), 10000 random walks x 1000 timepoints, every target"),
    "linf": ([os.path.join(root, "tools", "time_strict.py"), "1000000", "3", "walks"], "1000000x3",
             "strict L-infinity depth of 10^6 points in R^3 (state-class kernel)"),
    "medium": ([os.path.join(root, "tools", "time_rank.py"), "20000", "500", "10"], "20000x500",
               "medium route: 20000 curves x 500 timepoints through two column blocks per workgroup"),
    "calib": ([os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--algo",
               "pairwise", "--rotate", "1"], "10000x1000", "calibration: nan_count_rows_kernel reads 80 000 000 bytes once"),
}


def run(cmd, d):
    shutil.rmtree(d, ignore_errors=True)             # an earlier run's files under the same tag would be picked up below
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    print(" ".join(cmd[:6]), "... rc", r.returncode, flush=True)
    if r.returncode:
        print(r.stderr[-2000:])
    return r.returncode


def stats(name):
    d = os.path.join(out, f"{tag}_{name}_stats")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", PY] + WORKLOADS[name][0], d)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "sd::" in r["Name"]]
    dst = os.path.join(out, f"{tag}_{name}_kernel_stats.csv")
    with open(dst, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"].split("(")[0].replace("void ", ""), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["MinNs"], r["MaxNs"]])
    for r in rows:
        print(f'   {r["Name"].split("(")[0][:70]:70s} calls {r["Calls"]:>5} avg {float(r["AverageNs"]) / 1e3:9.2f} us')


def pmc(name):
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, f"{tag}_{name}_pmc_{counter}")
        run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", PY] + WORKLOADS[name][0], d)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for row in csv.DictReader(open(f)):
                if "sd::" not in row["Kernel_Name"] or row["Counter_Name"] != counter:
                    continue
                kn = row["Kernel_Name"].split("(")[0].replace("void ", "")
                a = acc.setdefault(kn, [0.0, set()])
                a[0] += float(row["Counter_Value"])
                a[1].add(row["Dispatch_Id"])
            for kn, (v, ds) in acc.items():
                res.setdefault(kn, {})[f"{counter}_KiB_per_launch"] = v / max(1, len(ds))
                res[kn]["launches"] = len(ds)
    for kn, d in res.items():
        d["hbm_bytes_per_launch_corrected"] = (2.0 * d.get("FETCH_SIZE_KiB_per_launch", 0.0) + d.get("WRITE_SIZE_KiB_per_launch", 0.0)) * 1024.0
    return res


only = sys.argv[2:]
for name in ("bench", "config3", "strict", "linf", "medium"):
    if not only or name in only:
        stats(name)
if os.environ.get("SD_STATS_ONLY"):                # experiments: the kernel durations only
    sys.exit(0)
calib = pmc("calib") if (not only or "bench" in only or "config3" in only) else {}
for name in ("bench", "config3"):
    if only and name not in only:
        continue
    ks = pmc(name)
    doc = {"workload": WORKLOADS[name][1], "command": " ".join(os.path.relpath(a, root) if os.path.isabs(a) else a for a in WORKLOADS[name][0]),
           "note": WORKLOADS[name][2] + ".  rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel trace only); values in KiB per "
                   "launch; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of a coalesced streaming "
                   "read, MI355X_MICROARCH.md HBM section), calibrated below on a kernel of known traffic.",
           "kernels": ks,
           "calibration": {k: v for k, v in calib.items() if "nan_count_rows" in k},
           "step_total_hbm_bytes_corrected": sum(v["hbm_bytes_per_launch_corrected"] for v in ks.values())}
    json.dump(doc, open(os.path.join(out, f"{tag}_{name}_pmc_traffic.json"), "w"), indent=1)
    print(name, "HBM bytes per step (corrected):", doc["step_total_hbm_bytes_corrected"])
    for kn, v in ks.items():
        print(f"   {kn[:70]:70s} {v['hbm_bytes_per_launch_corrected'] / 1e6:9.2f} MB")
