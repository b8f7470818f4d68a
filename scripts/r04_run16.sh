#!/bin/bash
# r04 GPU session 16: speculative frame loop over long tracks -- parity tests, then single-sequence latency (graph + eager)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_16; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_visual_prepare.py -m gpu -x -q -k "whole_frame_loop or contention or quota" 2>&1 | tail -15 | tee $O/tests.log
for ns in 0 1; do
  echo "no_speculation=$ns graph"; HV_EKF_NO_SPECULATION=$ns timeout 300 python scripts/lanes_probe.py 1 1 200 2>&1 | tail -1
  echo "no_speculation=$ns eager"; HV_EKF_NO_SPECULATION=$ns LP_EAGER=1 timeout 300 python scripts/lanes_probe.py 1 1 200 2>&1 | tail -1
done | tee $O/latency.txt
