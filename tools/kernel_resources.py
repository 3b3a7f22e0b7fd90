#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{root}/include",
       f"-I{root}/statdepth_amd/csrc", "-ffp-contract=off", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:") or txt.startswith("Name:"):
        cur = {"name": txt.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in txt:
        k, v = txt.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    print(f"{name:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} "
          f"spillV {r.get('VGPRs Spill','?'):>3} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")
