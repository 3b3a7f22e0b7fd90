#!/bin/bash
# exp_data.sh OUT: config-2 shape under the data variants of tools/time_rank.py (product library; 4 matrices in rotation)
out=$1; root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; : > $out
for env in "" "SD_TIES=1" "SD_CAUCHY=1" "SD_OUTLIER=3" "SD_OUTLIER_RANDOM=100" "SD_SORTED=normal" "SD_SORTED=t3"; do
  echo "== ${env:-walks}" >> $out
  env $env SD_ROTATE=4 timeout -k 10 120 python3 tools/time_rank.py ${N:-10000} ${T:-1000} ${REPS:-200} 2>&1 | grep -v amdgpu.ids >> $out || echo FAILED >> $out
done
cat $out
