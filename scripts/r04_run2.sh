#!/bin/bash
# r04 GPU session 2: lane-count / batch-size probes with the no-captured-fork default, fused-long A/B, kernel trace of two lanes
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_2; mkdir -p $O
export TMPDIR=/tmp
P() { timeout 400 python scripts/lanes_probe.py "$@" 2>&1 | grep -a "LANES_PROBE\|Error\|error" | tail -2 >> $O/probe.log; }
LP_PRELOAD=0 P 2 1024 24
LP_PRELOAD=6 P 2 1024 24
P 1 1024 24
HV_EKF_LONG_FUSED=0 P 1 1024 24
HV_EKF_LONG_FUSED=0 P 2 1024 24
P 3 1024 24
P 4 1024 24
P 2 1536 16
P 2 2048 12
P 4 512 24
cat $O/probe.log
bash scripts/trace_lanes.sh 2 1024 r04_2/trace_lanes2 2>&1 | tail -30
