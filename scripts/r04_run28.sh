#!/bin/bash
# r04 GPU session 28: the long build writes the compact Jacobian behind the gate, for inliers only -- parity tests, then 4 lanes against
# the library of the commit before (hybvio_amd/lib/libhybvio_hip_ab.so)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_28; mkdir -p $O
export TMPDIR=/tmp
OLD=$(pwd)/hybvio_amd/lib/libhybvio_hip_ab.so
timeout 600 python -m pytest tests/test_gpu_visual_prepare.py tests/test_gpu_lanes.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.log
echo "old 4x1024"; HV_LIB_OVERRIDE=$OLD timeout 200 python scripts/lanes_probe.py 4 1024 16 2>&1 | tail -1 | tee $O/ab4.txt
echo "new 4x1024"; timeout 200 python scripts/lanes_probe.py 4 1024 16 2>&1 | tail -1 | tee -a $O/ab4.txt
echo "new 1x1"; timeout 100 python scripts/lanes_probe.py 1 1 100 2>&1 | tail -1 | tee $O/latency.txt
