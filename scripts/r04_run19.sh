#!/bin/bash
# r04 GPU session 19: phase stamps (library built with HV_EKF_PHASE_STAMPS=1) of ONE track visit at B = 1: 21, 16, 13, 12 and 6 stereo poses
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_19; mkdir -p $O
export TMPDIR=/tmp
for np in 21 16 13 12 6; do
  echo "== poses $np"; HV_RAW_STAMPS=1 HV_EKF_PHASE_STAMPS=1 timeout 120 python scripts/vu_microbench.py 1 $np 1 2>&1 | grep -v amdgpu.ids
done | tee $O/stamps.txt
