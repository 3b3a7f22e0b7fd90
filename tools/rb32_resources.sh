#!/bin/bash
# registers / scratch of every rank_bucket32_kernel instantiation under extra flags (CPU only; hipcc cross-compiles)
cd "$(dirname "$0")/../statdepth_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function -ffp-contract=off "$@" \
  -Rpass-analysis=kernel-resource-usage -c mbd_rank_bucket32.hip -o /tmp/rb32_res.o 2>&1 | \
  grep -E "error|Function Name|VGPRs:|ScratchSize" | sed -e 's/.*Function Name: _ZN2sd20rank_bucket32_kernelILi\([0-9]*\)E.*/E=\1/' -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - 
