#!/bin/bash
# exp_batch.sh OUT VARIANT...: config-2 timings (4 matrices in rotation) of the product library and of each variant library
# statdepth_amd/lib/libsd_VARIANT.so (timing experiments, GPU box only; build the variants with tools/build_variant.sh)
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
: > $out
for v in product "$@"; do
  if [ "$v" = product ]; then unset SD_LIB; else export SD_LIB=$root/statdepth_amd/lib/libsd_$v.so; fi
  for shape in ${SHAPES:-10000:1000}; do
    n=${shape%%:*}; T=${shape##*:}
    echo "== $v n=$n T=$T" >> $out
    SD_ROTATE=4 timeout -k 10 120 python3 tools/time_rank.py $n $T ${REPS:-400} >> $out 2>&1 || echo "FAILED $v" >> $out
  done
done
cat $out
