#!/bin/bash
# build_variant.sh NAME FLAGS...  -> statdepth_amd/lib/libsd_NAME.so: the product objects with mbd_rank_big.hip (and, if
# named in SRC, another source) recompiled under FLAGS (timing experiments; select with SD_LIB=... tools/time_rank.py)
set -e
name=$1; shift
src=${SRC:-mbd_rank_big}
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/statdepth_amd/csrc
mkdir -p ../lib/obj_var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I. -Wall -Wno-unused-function -ffp-contract=off "$@" -c $src.hip -o ../lib/obj_var/${src}_$name.o
objs=""
for o in sd_api mbd_pairwise mbd_rank_bucket mbd_rank_bucket32 bd_strict bd_strict_grid l1_depth simplex band_enum xcheck mbd_rank_ab mbd_rank_big; do
  if [ "$o" = "$src" ]; then objs="$objs ../lib/obj_var/${src}_$name.o"; else objs="$objs ../lib/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libsd_$name.so $objs
echo built ../lib/libsd_$name.so
