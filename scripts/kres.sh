#!/bin/bash
# developer aid: register / spill report of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), asm kept in /tmp
# usage: scripts/kres.sh ekf.hip [name-filter]
cd "$(dirname "$0")/../hybvio_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c "$1" -o /tmp/kres_test.o -Rpass-analysis=kernel-resource-usage -save-temps=obj 2>&1 \
  | grep -A12 "Function Name.*${2:-}\|error" | grep -i "error\|Function Name\|VGPRs:\|Spill\|Scratch\|Occupancy" | sed 's/\[-Rpass.*//; s/^.*remark: *//'
