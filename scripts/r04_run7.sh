#!/bin/bash
# r04 GPU session 7: schedule A/B of the captured one-stream lane visit (4 x 1024), new lane tests
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_7; mkdir -p $O
export TMPDIR=/tmp
P() { timeout 400 python scripts/lanes_probe.py "$@" 2>&1 | grep -a "LANES_PROBE\|Error\|error" | tail -2 >> $O/probe.log; }
P 4 1024 16
HV_EKF_LONG_FIRST=0 P 4 1024 16
HV_EKF_DUAL_UPDATE=0 P 4 1024 16
HV_EKF_VISIT_ORDER=0 P 4 1024 16
HV_EKF_LONG_FIRST=0 P 2 1024 24
P 6 1024 12
cat $O/probe.log
( time timeout 600 python -m pytest tests/test_gpu_lanes.py -q --maxfail=5 ) > $O/tests_lanes.log 2>&1
tail -5 $O/tests_lanes.log
