#!/bin/bash
# r04 GPU session 22: as session 21 with the factor form in the LONG build only (vu_gate_long_kernel); the short class keeps the MFMA products
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_22; mkdir -p $O
export TMPDIR=/tmp
OLD=$(pwd)/hybvio_amd/lib/libhybvio_hip_ab.so
timeout 1200 python -m pytest tests/test_gpu_visual_prepare.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.log
for r in 1 2; do
  echo "old 4x1024"; HV_LIB_OVERRIDE=$OLD timeout 300 python scripts/lanes_probe.py 4 1024 24 2>&1 | tail -1
  echo "new 4x1024"; timeout 300 python scripts/lanes_probe.py 4 1024 24 2>&1 | tail -1
done | tee $O/ab4.txt
for r in 1 2; do
echo "old 1x1024 eager"; HV_LIB_OVERRIDE=$OLD LP_EAGER=1 timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | tail -1
echo "new 1x1024 eager"; LP_EAGER=1 timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | tail -1
done | tee $O/ab1.txt
echo "new 1x1"; timeout 300 python scripts/lanes_probe.py 1 1 200 2>&1 | tail -1 | tee $O/ab_latency.txt
