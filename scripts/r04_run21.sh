#!/bin/bash
# r04 GPU session 21: A/B on ONE box -- the fused gates on the Jacobian's factors (structured_S, working tree) against the previous commit's
# dense MFMA products (hybvio_amd/lib/libhybvio_hip_ab.so, built from `git archive HEAD`): parity tests, then 4 lanes / 1 lane / 1 sequence
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_21; mkdir -p $O
export TMPDIR=/tmp
OLD=$(pwd)/hybvio_amd/lib/libhybvio_hip_ab.so
timeout 1200 python -m pytest tests/test_gpu_visual_prepare.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.log
for r in 1 2; do
  echo "old 4x1024"; HV_LIB_OVERRIDE=$OLD timeout 300 python scripts/lanes_probe.py 4 1024 24 2>&1 | tail -1
  echo "new 4x1024"; timeout 300 python scripts/lanes_probe.py 4 1024 24 2>&1 | tail -1
done | tee $O/ab4.txt
echo "old 1x1024 eager"; HV_LIB_OVERRIDE=$OLD LP_EAGER=1 timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | tail -1 | tee $O/ab1.txt
echo "new 1x1024 eager"; LP_EAGER=1 timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | tail -1 | tee -a $O/ab1.txt
echo "old 1x1"; HV_LIB_OVERRIDE=$OLD timeout 300 python scripts/lanes_probe.py 1 1 200 2>&1 | tail -1 | tee $O/ab_latency.txt
echo "new 1x1"; timeout 300 python scripts/lanes_probe.py 1 1 200 2>&1 | tail -1 | tee -a $O/ab_latency.txt
for np in 21 16 10; do
  echo "old B=1024 np=$np"; HV_LIB_OVERRIDE=$OLD timeout 120 python scripts/vu_microbench.py 1024 $np 1 2>&1 | grep "all rejected"
  echo "new B=1024 np=$np"; timeout 120 python scripts/vu_microbench.py 1024 $np 1 2>&1 | grep "all rejected"
done | tee $O/ab_visit.txt
