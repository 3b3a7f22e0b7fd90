#!/bin/bash
# exp_rpb.sh OUT: the large-n route with fewer rows per batch (cross-check library, SD_RANK_ROWS_PER_BATCH), so that a batch's
# records (8 bytes per key between the partition and the ranking kernel) are still in the 256 MB Infinity Cache when they are
# read back.  Timing experiment, GPU box only.
out=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
: > $out
export SD_LIB=$root/statdepth_amd/lib/libstatdepth_hip_xcheck.so
for shape in ${SHAPES:-100000:256 100000:1000}; do
  n=${shape%%:*}; T=${shape##*:}
  for rpb in ${RPBS:-0 128 64 32 16}; do
    if [ "$rpb" = 0 ]; then unset SD_RANK_ROWS_PER_BATCH; else export SD_RANK_ROWS_PER_BATCH=$rpb; fi
    echo "== rows per batch $rpb (0: the plan's own) n=$n T=$T" >> $out
    SD_ROTATE=${ROT:-3} timeout -k 10 200 python3 tools/time_rank.py $n $T ${REPS:-60} >> $out 2>&1 || echo "FAILED rpb=$rpb" >> $out
  done
done
cat $out
