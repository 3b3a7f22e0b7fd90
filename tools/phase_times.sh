#!/bin/bash
# phase_times.sh KERNEL n T VARIANT...: average duration of KERNEL (rocprofv3 --kernel-trace --stats) under the product
# library and under each variant library statdepth_amd/lib/libsd_VARIANT.so (timing experiments, GPU box only)
kern=$1; n=$2; T=$3; shift 3
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for v in product "$@"; do
  d=$root/gpurun_out/phase_$v
  rm -rf $d
  if [ "$v" = product ]; then unset SD_LIB; else export SD_LIB=$root/statdepth_amd/lib/libsd_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/time_rank.py $n $T 5 > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" "$kern" <<'PY'
import csv, re, sys
v, f, kern = sys.argv[1:4]
out = []
for r in csv.DictReader(open(f)):
    if re.search(kern, r["Name"]):
        out.append(f'{r["Name"].split("(")[0].replace("void ", "").replace("sd::", "")} {float(r["AverageNs"]) / 1e3:.2f} us')
print(f"{v}: " + "; ".join(sorted(out)), flush=True)
PY
  rm -rf $d
done
