#!/bin/bash
# r04 GPU session 25 (experiment, NOT in the tree): LDS-only barriers in the update kernel's LDS phases, so that the P tiles requested ahead
# stay in flight across the staging barriers -- full GPU suite (318 passed), then A/B against the previous commit's library on the same
# box: no difference (update launch 55.7 / 55.9 us at 25 % inliers, 203.6 / 207.9 all inliers; 4 lanes 140.4 / 140.4 k): reverted
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_25; mkdir -p $O
export TMPDIR=/tmp
OLD=$(pwd)/hybvio_amd/lib/libhybvio_hip_ab.so
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.log
for r in 1 2; do
  echo "old 4x1024"; HV_LIB_OVERRIDE=$OLD timeout 300 python scripts/lanes_probe.py 4 1024 24 2>&1 | tail -1
  echo "new 4x1024"; timeout 300 python scripts/lanes_probe.py 4 1024 24 2>&1 | tail -1
done | tee $O/ab4.txt
echo "old 1x1024 eager"; HV_LIB_OVERRIDE=$OLD LP_EAGER=1 timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | tail -1 | tee $O/ab1.txt
echo "new 1x1024 eager"; LP_EAGER=1 timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | tail -1 | tee -a $O/ab1.txt
echo "old B=1024 np=10"; HV_LIB_OVERRIDE=$OLD timeout 120 python scripts/vu_microbench.py 1024 10 1 2>&1 | grep "all inliers\|0.25" | tee $O/ab_update.txt
echo "new B=1024 np=10"; timeout 120 python scripts/vu_microbench.py 1024 10 1 2>&1 | grep "all inliers\|0.25" | tee -a $O/ab_update.txt
