#!/bin/bash
# r04 GPU session 9: forked visit with the roles of the two streams swapped (knob ekf_side_stream 6) against r03's arrangement (3)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_9; mkdir -p $O
export TMPDIR=/tmp
P() { timeout 400 python scripts/lanes_probe.py "$@" 2>&1 | grep -a "LANES_PROBE\|Error\|error" | tail -2 >> $O/probe.log; }
LP_MODE=torch HV_EKF_SIDE_STREAM=3 P 1 1024 24
LP_MODE=torch HV_EKF_SIDE_STREAM=6 P 1 1024 24
LP_MODE=torch HV_EKF_SIDE_STREAM=0 P 1 1024 24
LP_MODE=torch LP_EAGER=1 HV_EKF_SIDE_STREAM=3 P 1 1024 24
LP_MODE=torch LP_EAGER=1 HV_EKF_SIDE_STREAM=6 P 1 1024 24
LP_MODE=lanes LP_EAGER=1 P 1 1024 24
LP_MODE=lanes LP_EAGER=1 HV_EKF_SIDE_STREAM=6 P 1 1024 24
LP_MODE=lanes HV_EKF_SIDE_STREAM=6 P 2 1024 24
cat $O/probe.log
